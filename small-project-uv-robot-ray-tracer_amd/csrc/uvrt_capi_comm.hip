// uvrt_capi_comm.hip -- the one collective of a sharded computation (RCCL over xGMI)
// (the C ABI of include/uvrt.h over the HIP kernels; the context and its helpers are in uvrt_ctx.h)
#include "uvrt_ctx.h"

#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl is opened at run time

using namespace uvrt;
using namespace uvrt_impl;

// librccl is opened lazily with dlopen: a process that never shards (the common case) does not load it,
// and one that already holds an RCCL (torch.distributed) gets that same library by its soname.
namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

int rccl_load()
{
    if (g_rccl.lib) return UVRT_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) return fail(UVRT_ERR_HIP, "uvrt_comm: cannot open librccl (%s)", dlerror());
#define UVRT_SYM(field, sym)                                                                   \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, #sym);                                      \
    if (!g_rccl.field) return fail(UVRT_ERR_HIP, "uvrt_comm: librccl lacks " #sym)
    UVRT_SYM(GetUniqueId, ncclGetUniqueId);
    UVRT_SYM(CommInitRank, ncclCommInitRank);
    UVRT_SYM(CommInitAll, ncclCommInitAll);
    UVRT_SYM(CommDestroy, ncclCommDestroy);
    UVRT_SYM(CommCount, ncclCommCount);
    UVRT_SYM(AllReduce, ncclAllReduce);
    UVRT_SYM(GroupStart, ncclGroupStart);
    UVRT_SYM(GroupEnd, ncclGroupEnd);
    UVRT_SYM(GetErrorString, ncclGetErrorString);
#undef UVRT_SYM
    g_rccl.lib = h;
    return UVRT_OK;
}
}  // namespace

namespace uvrt_impl {

// RCCL's device kernels on gfx950 (ncclDevKernel_Generic_*, librccl 2.x of ROCm 7.2: read off the code object's kernel
// descriptors) take 37 664 B of LDS and 248-256 VGPRs per wave in workgroups of up to 512 threads: such a workgroup
// cannot become resident on a CU that holds more than four of k_extend6's (20 KB of LDS, 64 VGPRs), and the persistent
// tracing grid holds seven on EVERY CU until its launch drains -- the all-reduce of batch k would wait for the trace of
// batch k + 1 to end instead of running beside it.  So while a communicator is set, the launch lanes' streams are created
// with a CU mask (hipExtStreamCreateWithCUMask) that leaves `reserve` CUs to the context's stream, where the fold, the
// collective and the replay run.  On the MI355X mask bit b is XCC b % 8 (tests/tools/cumask_probe.hip,
// profiles/r03/r03_cumask_probe.txt): clearing the LAST eight bits takes one CU from every XCD, so the round-robin deal of
// workgroups to XCDs stays balanced.
int set_lane_cu_mask(uvrt_ctx* c, int reserve)
{
    // the mask layout was probed on the whole MI355X (256 CUs, 8 XCDs); a partitioned device keeps its plain streams
    if (reserve < 0 || reserve % 8 != 0 || c->num_cus != 256) reserve = 0;
    if (reserve == c->lanes_masked_cus) return UVRT_OK;
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t words = (uint32_t)((c->num_cus + 31) / 32);
    std::vector<uint32_t> mask(words, 0xFFFFFFFFu);
    for (int k = 0; k < reserve; ++k) { const int bit = c->num_cus - 1 - k; mask[bit / 32] &= ~(1u << (bit % 32)); }
    // every new stream is created before any lane is touched: a failure midway leaves the lanes as they were.
    // hipExtStreamCreateWithCUMask takes no flags: masked lanes are BLOCKING streams, i.e. they synchronise implicitly with
    // the legacy null stream.  The library itself never uses the null stream; a host application that does (synchronous
    // hipMemcpy, default-stream kernels) serialises with the tracing lanes while a communicator is set.
    hipStream_t fresh[uvrt_ctx::MAXL] = {};
    for (int l = 1; l < uvrt_ctx::MAXL; ++l) {
        hipError_t e = reserve > 0 ? hipExtStreamCreateWithCUMask(&fresh[l], words, mask.data())
                                   : hipStreamCreateWithFlags(&fresh[l], hipStreamNonBlocking);
        if (e != hipSuccess) {
            for (int k = 1; k < l; ++k) (void)hipStreamDestroy(fresh[k]);
            return fail(UVRT_ERR_HIP, "uvrt_comm: cannot create launch-lane stream %d (%s); the lanes keep their streams", l, hipGetErrorString(e));
        }
    }
    for (int l = 1; l < uvrt_ctx::MAXL; ++l) {
        if (c->side[l]) { (void)hipStreamSynchronize(c->side[l]); (void)hipStreamDestroy(c->side[l]); }
        c->side[l] = fresh[l];
        c->side_used[l] = false;
        c->side_seen_fence[l] = 0;            // the new stream has seen no fence: it waits for the current ones at first use
        c->side_seen_mapfence[l] = 0;
    }
    c->lanes_masked_cus = reserve;
    return UVRT_OK;
}

}  // namespace uvrt_impl

namespace {
#define RCCL_TRY(expr)                                                                         \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(UVRT_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));        \
    } while (0)
}  // namespace

extern "C" {

int uvrt_comm_unique_id(void* id128)
{
    if (!id128) return fail(UVRT_ERR_INVALID, "uvrt_comm_unique_id: null pointer");
    if (int rc = rccl_load()) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id over as 128 bytes");
    RCCL_TRY(g_rccl.GetUniqueId((ncclUniqueId*)id128));
    return UVRT_OK;
}

int uvrt_comm_available(void)
{
    return rccl_load() == UVRT_OK ? 1 : 0;
}

int uvrt_comm_info(uvrt_ctx* c, int32_t out4[4])
{
    if (!c || !out4) return fail(UVRT_ERR_INVALID, "uvrt_comm_info: null pointer");
    out4[0] = c->comm ? c->comm_world : 0;
    out4[1] = c->comm ? c->comm_rank : 0;
    out4[2] = 0;
    out4[3] = c->lanes_masked_cus;
    if (c->comm) {
        int n = 0;
        RCCL_TRY(g_rccl.CommCount((ncclComm_t)c->comm, &n));      // what RCCL itself says the communicator spans
        out4[2] = n;
    }
    return UVRT_OK;
}

int uvrt_comm_init_rank(uvrt_ctx* c, const void* id128, int32_t rank, int32_t world)
{
    if (!c || !id128 || world < 1 || rank < 0 || rank >= world) return fail(UVRT_ERR_INVALID, "uvrt_comm_init_rank: bad argument");
    if (c->comm) return fail(UVRT_ERR_INVALID, "uvrt_comm_init_rank: the context already has a communicator");
    if (int rc = rccl_load()) return rc;
    if (int rc = set_device(c)) return rc;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    return set_lane_cu_mask(c, c->comm_reserve_knob);
}

int uvrt_comm_init_all(uvrt_ctx** ctxs, int32_t n)
{
    if (!ctxs || n < 1 || n > 64) return fail(UVRT_ERR_INVALID, "uvrt_comm_init_all: bad argument");
    int devs[64];
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->comm) return fail(UVRT_ERR_INVALID, "uvrt_comm_init_all: null context or communicator present");
        devs[i] = ctxs[i]->device;
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i])
                return fail(UVRT_ERR_INVALID, "uvrt_comm_init_all: contexts %d and %d share device %d (RCCL wants one rank "
                            "per device; uvrt_reduce_batch_group sums contexts of one device without it)", j, i, devs[i]);
    }
    if (int rc = rccl_load()) return rc;
    ncclComm_t comms[64];
    RCCL_TRY(g_rccl.CommInitAll(comms, n, devs));
    for (int i = 0; i < n; ++i) { ctxs[i]->comm = comms[i]; ctxs[i]->comm_rank = i; ctxs[i]->comm_world = n; }
    for (int i = 0; i < n; ++i)
        if (int rc = set_lane_cu_mask(ctxs[i], ctxs[i]->comm_reserve_knob)) return rc;
    return UVRT_OK;
}

int uvrt_comm_destroy(uvrt_ctx* c)
{
    if (!c || !c->comm) return UVRT_OK;
    if (g_rccl.lib) {
        (void)hipSetDevice(c->device);
        (void)hipStreamSynchronize(c->stream);
        (void)g_rccl.CommDestroy((ncclComm_t)c->comm);
    }
    c->comm = nullptr;
    c->comm_world = 1;
    c->comm_rank = 0;
    return set_lane_cu_mask(c, 0);
}

int uvrt_reduce_batch(uvrt_ctx* c)
{
    if (!c || c->b_count <= 0) return fail(UVRT_ERR_INVALID, "uvrt_reduce_batch: no traced batch");
    if (!c->comm) return fail(UVRT_ERR_INVALID, "uvrt_reduce_batch: no communicator (uvrt_comm_init_rank / uvrt_comm_init_all)");
    if (int rc = uvrt_fold_batch(c)) return rc;
    if (int rc = set_device(c)) return rc;
    RCCL_TRY(g_rccl.AllReduce(c->bs[c->b_set].folded.p, c->bs[c->b_set].folded.p, (size_t)c->b_count * (size_t)c->T, ncclInt32, ncclSum,
                              (ncclComm_t)c->comm, c->stream));
    return UVRT_OK;
}

int uvrt_reduce_batch_group(uvrt_ctx** ctxs, int32_t n)
{
    if (!ctxs || n < 1) return fail(UVRT_ERR_INVALID, "uvrt_reduce_batch_group: bad argument");
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i] || ctxs[i]->b_count <= 0 || ctxs[i]->b_count != ctxs[0]->b_count || ctxs[i]->T != ctxs[0]->T)
            return fail(UVRT_ERR_INVALID, "uvrt_reduce_batch_group: context %d holds no batch of the same shape", i);
        if (int rc = uvrt_fold_batch(ctxs[i])) return rc;
    }
    if (n == 1) return UVRT_OK;
    const size_t count = (size_t)ctxs[0]->b_count * (size_t)ctxs[0]->T;
    if (ctxs[0]->comm) {          // one process, one device per context: a grouped RCCL all-reduce
        if (int rc = rccl_load()) return rc;
        RCCL_TRY(g_rccl.GroupStart());
        for (int i = 0; i < n; ++i) {
            if (!ctxs[i]->comm) { (void)g_rccl.GroupEnd(); return fail(UVRT_ERR_INVALID, "uvrt_reduce_batch_group: context %d has no communicator", i); }
            HIP_TRY(hipSetDevice(ctxs[i]->device));
            RCCL_TRY(g_rccl.AllReduce(ctxs[i]->bs[ctxs[i]->b_set].folded.p, ctxs[i]->bs[ctxs[i]->b_set].folded.p, count, ncclInt32, ncclSum,
                                      (ncclComm_t)ctxs[i]->comm, ctxs[i]->stream));
        }
        RCCL_TRY(g_rccl.GroupEnd());
        return UVRT_OK;
    }
    // contexts of ONE device (rehearsals, tests): sum on context 0's stream, hand the result to the others
    for (int i = 1; i < n; ++i)
        if (ctxs[i]->device != ctxs[0]->device)
            return fail(UVRT_ERR_INVALID, "uvrt_reduce_batch_group: contexts on different devices need uvrt_comm_init_all first");
    uvrt_ctx* c0 = ctxs[0];
    if (int rc = set_device(c0)) return rc;
    for (int i = 1; i < n; ++i) {
        HIP_TRY(hipEventRecord(ctxs[i]->ev_tail[0], ctxs[i]->stream));
        HIP_TRY(hipStreamWaitEvent(c0->stream, ctxs[i]->ev_tail[0], 0));
        launch_add_counts(c0->bs[c0->b_set].folded.as<int32_t>(), ctxs[i]->bs[ctxs[i]->b_set].folded.as<int32_t>(), (int64_t)count, c0->stream);
    }
    HIP_TRY(hipGetLastError());
    for (int i = 1; i < n; ++i)
        HIP_TRY(hipMemcpyAsync(ctxs[i]->bs[ctxs[i]->b_set].folded.p, c0->bs[c0->b_set].folded.p, count * 4, hipMemcpyDeviceToDevice, c0->stream));
    HIP_TRY(hipEventRecord(c0->ev_tail[0], c0->stream));
    for (int i = 1; i < n; ++i) {
        HIP_TRY(hipStreamWaitEvent(ctxs[i]->stream, c0->ev_tail[0], 0));
    }
    return UVRT_OK;
}

}  // extern "C"
