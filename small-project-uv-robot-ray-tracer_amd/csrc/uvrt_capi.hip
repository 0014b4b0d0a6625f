// uvrt_capi.hip -- the C ABI of include/uvrt.h over the HIP kernels.
//
// One context = one HIP device + one in-order stream + all device buffers of a RayTracer
// (raytracer.h:50-53).  There is no CPU fallback: every entry point either runs on the GPU or
// returns an error.
#include "../../include/uvrt.h"
#include "uvrt_device.h"

#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl is opened at run time (uvrt_comm_*)

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace uvrt;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(UVRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                   \
                        hipGetErrorString(e_), __FILE__, __LINE__);                     \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    // Zeroing is enqueued on `s`, the stream every kernel of the context runs on (the
    // context's stream is non-blocking, so a null-stream hipMemset would not be ordered
    // against it).
    int ensure(size_t need, bool zero, hipStream_t s)
    {
        if (need <= bytes && p) return UVRT_OK;
        if (p) { HIP_TRY(hipFree(p)); p = nullptr; bytes = 0; }
        if (need == 0) return UVRT_OK;
        HIP_TRY(hipMalloc(&p, need));
        bytes = need;
        if (zero) HIP_TRY(hipMemsetAsync(p, 0, need, s));
        return UVRT_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

}  // namespace

struct uvrt_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    // scene
    int32_t T = 0;
    DevBuf pairs, recs, perm, ltris, leaf_count, area;
    bool have_perm = false;      // the caller's own record renumbering (uvrt_set_record_perm)
    int32_t npairs = 0;
    uint32_t root_ref = REF_DONE;
    uint32_t top_pairs = 0;      // inner nodes of the first 7 tree levels (breadth-first prefix of `pairs`)
    bool have_scene = false;
    int32_t replicas = 1;        // deposit replicas of tempPhotonMap (uvrt_device.h ExtendParams)
    int32_t replicas_knob = -1;  // -1: choose from T

    // per-triangle maps (raytracer.cpp:32-37)
    DevBuf photon_map, max_map, counts, dosage, color;

    // rays
    int64_t capacity = 0;
    DevBuf rays, keyrank, sorted, order, hits, hist, bin_start, export_buf;
    DevBuf ovf_stack;                          // traversal-stack entries 8..31 of every thread of the persistent grid
    bool recs_valid = false;                   // recs[0, npairs) prepared for the lamp (recs_ox, recs_oz)
    float recs_ox = 0, recs_oz = 0;
    bool scene_force_exact = false;            // a node bound too tiny / too large for the reciprocal shortcuts
    int32_t hist_bins = 0;
    int64_t last_n = -1;
    int64_t last_first = 0;
    bool last_sorted = false;
    bool last_extended = false;
    float ox = 0, oz = 0;

    // Launch lanes: consecutive launches (generate -> extend -> accumulate ->
    // shade) alternate between the context's stream and an internal side stream, each with its own
    // ray, record, count and overflow-stack buffers, so that the next launch fills the wave slots the
    // draining launch frees.  The per-triangle maps are updated in launch order (event waits).
    static constexpr int MAXL = 4;    // lane 0 = the context's stream and the buffers above
    bool pipeline = true;             // uvrt_set_pipeline
    int nlanes = 2;                   // developer knob UVRT_LANES (1..MAXL): 3 gain ~1 %, 4 (with 4 workgroups
                                      // per CU) win only for long launch sequences (profiles/r01_v6_experiments.txt)
    bool ext_touch = false;           // a count-buffer pointer was handed out since the last fence
    bool ext_touch_maps = false;      // a map / dose / colour pointer was handed out since the last map fence
    bool counts_dirty[MAXL] = {};     // the lane's count buffer holds deposits that were not accumulated
    hipEvent_t ev_mapfence = nullptr; // on the main stream, after the last operation on the per-triangle maps
    uint64_t mapfence_seq = 0, side_seen_mapfence[MAXL] = {};
    int lane = 0;                     // lane of the current launch (uvrt_generate selects it)
    int prev_lane = 0;                // lane of the launch before it (the maps are updated in launch order)
    bool cur_pipelined = false;       // the current launch takes part in the lane rotation
    hipStream_t side[MAXL] = {};      // [0] unused
    bool side_used[MAXL] = {};        // the side stream holds work the main stream is not ordered after
    hipEvent_t ev_fence = nullptr;    // on the main stream, after the last context-wide operation
    hipEvent_t ev_tail[MAXL] = {};    // tail of a lane's stream
    uint64_t fence_seq = 0, side_seen_fence[MAXL] = {};
    DevBuf xrays[MAXL], xrecs[MAXL], xcounts[MAXL], xovf[MAXL];   // [0] unused: lane 0 has rays, recs, counts, ovf_stack
    bool xrecs_valid[MAXL] = {};
    float xrecs_ox[MAXL] = {}, xrecs_oz[MAXL] = {};

    // Opt-in 4-wide collapse of the BVH (uvrt_set_wide_bvh, uvrt_extend4.hip)
    bool wide = false;
    DevBuf quads;                         // [nquads] QuadRec, scene form
    DevBuf recs4[MAXL];                   // per lane: [2 * nquads + T + 1] 64-byte units, per-launch form + leaf records
    int32_t nquads = 0;
    uint32_t top_quads = 0;
    bool recs4_valid[MAXL] = {};          // recs4[l] hold the per-launch records of lamp column (recs4_ox, recs4_oz)
    float recs4_ox[MAXL] = {}, recs4_oz[MAXL] = {};

    // Hot-record renumbering per lamp position (uvrt_hotset.hip): the records a lamp's photons visit most are
    // the ones the traversal serves from LDS.  Built on the device the first time a lamp is seen.
    struct HotEntry { float lamp[3]; DevBuf perm, hist; uint64_t stamp; hipEvent_t ready; };
    std::vector<HotEntry> hot;
    uint64_t hot_clock = 0;
    int32_t hot_mode = 1;                 // uvrt_set_hot_records: 1 = automatic (default), 0 = breadth-first order
    const uint32_t* lane_perm[MAXL] = {}; // renumbering of the current launch of each lane (set by uvrt_generate)

    // Batched tracing (uvrt_trace_batch): the rays of up to MAX_BATCH launches side by side, one count
    // "plane" (replicas x T ints) per launch, one per-launch record array per distinct lamp.
    // two buffer sets: batch k + 1 is traced (on the launch lanes) into one while batch k is folded, reduced and
    // replayed (on the context's stream) out of the other
    struct BatchSet { DevBuf rays, planes, folded; hipEvent_t free_ev = nullptr; };
    BatchSet bs[2];
    int b_set = 0;                        // the set of the traced batch (b_count > 0) / of the last one
    uint64_t b_chunks = 0;                // chunks traced so far: consecutive chunks alternate over the launch lanes
    int32_t b_repl = 16;                  // deposit replicas per plane of the traced batch
    std::vector<DevBuf> b_recs;           // [group]
    struct RecsKey { float ox = 0, oz = 0; const uint32_t* perm = nullptr; bool valid = false; };
    std::vector<RecsKey> b_recs_key;      // what b_recs[g] holds
    int32_t b_count = 0;                  // launches of the batch that has not been replayed (0: none)
    int64_t b_n = 0, b_npad = 0;
    int32_t b_phys[MAX_BATCH] = {};       // logical launch -> physical plane (launches are grouped by lamp)
    bool b_is_folded = false;             // b_folded holds the batch (fold / all-reduce done), the replicas are zero
    void* comm = nullptr;                 // ncclComm_t of a ray-range-sharded job (uvrt_comm_init_rank)
    int comm_rank = 0, comm_world = 1;

    // generate.cl:6 program-scope SEED
    uint32_t seed = 0;
    int32_t seed_mode = 0;   // uvrt_set_seed_mode

    // knobs
    int32_t sort_bits = 0;   // ray ordering off by default: extend is VALU-bound
    bool record_hits = false;
    int32_t variant = 0;
    int32_t flavour = 0;

    // traversal error flag + extend timing
    DevBuf error_flag;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
};

namespace {

int set_device(uvrt_ctx* c)
{
    HIP_TRY(hipSetDevice(c->device));
    return UVRT_OK;
}

// ---- launch lanes ----
hipStream_t stream_of(uvrt_ctx* c, int l) { return l == 0 ? c->stream : c->side[l]; }
// the main stream becomes ordered after everything the side streams hold
int join_all(uvrt_ctx* c)
{
    for (int l = 1; l < uvrt_ctx::MAXL; ++l) {
        if (!c->side_used[l]) continue;
        HIP_TRY(hipEventRecord(c->ev_tail[l], c->side[l]));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_tail[l], 0));
        c->side_used[l] = false;
    }
    return UVRT_OK;
}
// a context-wide operation has been enqueued on the main stream: later side-stream work waits for it
int mark_fence(uvrt_ctx* c)
{
    HIP_TRY(hipEventRecord(c->ev_fence, c->stream));
    ++c->fence_seq;
    return UVRT_OK;
}
// an operation on the per-triangle maps (reset, an external reduction) has been enqueued on the main
// stream: later accumulate / Shade work on side streams waits for it -- generate and extend do not,
// so the first launches of the next computation overlap the drain of the previous one
int mark_map_fence(uvrt_ctx* c)
{
    HIP_TRY(hipEventRecord(c->ev_mapfence, c->stream));
    ++c->mapfence_seq;
    return UVRT_OK;
}
// stream of the current lane; a side stream first catches up with the last context-wide operation
// and, for work on the maps (`maps`), with the last operation on them
int lane_stream(uvrt_ctx* c, hipStream_t* out, bool maps = false)
{
    // external work enqueued on the main stream since a device pointer was handed out
    if (c->ext_touch) { c->ext_touch = false; if (int rc = mark_fence(c)) return rc; }
    if (c->ext_touch_maps) { c->ext_touch_maps = false; if (int rc = mark_map_fence(c)) return rc; }
    const int l = c->lane;
    if (l == 0) { *out = c->stream; return UVRT_OK; }
    if (c->fence_seq != c->side_seen_fence[l]) {
        HIP_TRY(hipStreamWaitEvent(c->side[l], c->ev_fence, 0));
        c->side_seen_fence[l] = c->fence_seq;
    }
    if (maps && c->mapfence_seq != c->side_seen_mapfence[l]) {
        HIP_TRY(hipStreamWaitEvent(c->side[l], c->ev_mapfence, 0));
        c->side_seen_mapfence[l] = c->mapfence_seq;
    }
    c->side_used[l] = true;
    *out = c->side[l];
    return UVRT_OK;
}
// the current lane's stream becomes ordered after everything the previous launch's lane holds (its
// accumulate and Shade): the per-triangle maps are updated in launch order
int order_after_previous(uvrt_ctx* c)
{
    const int l = c->lane, q = c->prev_lane;
    if (q == l) return UVRT_OK;
    if (l == 0) return join_all(c);
    HIP_TRY(hipEventRecord(c->ev_tail[q], stream_of(c, q)));
    HIP_TRY(hipStreamWaitEvent(c->side[l], c->ev_tail[q], 0));
    c->side_used[l] = true;
    return UVRT_OK;
}
DevBuf& lane_rays(uvrt_ctx* c) { return c->lane ? c->xrays[c->lane] : c->rays; }
DevBuf& lane_recs(uvrt_ctx* c) { return c->lane ? c->xrecs[c->lane] : c->recs; }
DevBuf& lane_counts(uvrt_ctx* c) { return c->lane ? c->xcounts[c->lane] : c->counts; }
DevBuf& lane_ovf(uvrt_ctx* c) { return c->lane ? c->xovf[c->lane] : c->ovf_stack; }

// The record renumbering for a launch from `lamp`: the caller's own (uvrt_set_record_perm), the automatic
// hot-record one (statistics + selection enqueued on `s` the first time the lamp is seen), or none.
int launch_perm(uvrt_ctx* c, const float lamp[3], float light_length, uint32_t seed_prev, uint32_t seed_next, int64_t n,
                hipStream_t s, const uint32_t** out)
{
    *out = nullptr;
    if (c->have_perm) { *out = c->perm.as<uint32_t>(); return UVRT_OK; }
    if (c->hot_mode == 0 || c->npairs <= (int32_t)128 || c->root_ref >= REF_LEAF_BIT) return UVRT_OK;
    ++c->hot_clock;
    for (auto& h : c->hot)
        if (memcmp(h.lamp, lamp, 12) == 0) {
            h.stamp = c->hot_clock;
            HIP_TRY(hipStreamWaitEvent(s, h.ready, 0));      // it may have been built on another lane's stream
            *out = h.perm.as<uint32_t>();
            return UVRT_OK;
        }
    uvrt_ctx::HotEntry* e = nullptr;
    if (c->hot.size() < 64) {
        c->hot.emplace_back();
        e = &c->hot.back();
        if (int rc = e->perm.ensure((size_t)c->npairs * 4, false, s)) { c->hot.pop_back(); return rc; }
        if (int rc = e->hist.ensure((size_t)c->npairs * 4, true, s)) { e->perm.release(); c->hot.pop_back(); return rc; }
        HIP_TRY(hipEventCreateWithFlags(&e->ready, hipEventDisableTiming));
    } else {          // recycle the least recently used entry: nothing in flight may still read its renumbering
        if (int rc = join_all(c)) return rc;
        HIP_TRY(hipStreamSynchronize(c->stream));
        e = &c->hot[0];
        for (auto& h : c->hot) if (h.stamp < e->stamp) e = &h;
    }
    memcpy(e->lamp, lamp, 12);
    e->stamp = c->hot_clock;
    SceneDev sc;
    sc.pairs = c->pairs.as<PairRec>();
    sc.ltris = c->ltris.as<LeafTri>();
    sc.leaf_count = c->leaf_count.as<uint32_t>();
    sc.root_ref = c->root_ref;
    sc.tri_count = c->T;
    launch_visit_stats(sc, e->hist.as<uint32_t>(), lamp, light_length, seed_prev, seed_next, c->seed_mode,
                       (int32_t)std::min<int64_t>(n, 32768), s);
    launch_select_hot(e->hist.as<uint32_t>(), e->perm.as<uint32_t>(), c->npairs, (int32_t)TOP6_MAX, s);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e->ready, s));
    *out = e->perm.as<uint32_t>();
    return UVRT_OK;
}

// work-item 0's RNG walk of cl/generate.cl:13-39 on the host (strict f32/f64, same order)
uint32_t host_wang_hash(uint32_t s)
{
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    return s;
}
float host_random_float(uint32_t& s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return (float)s * 2.3283064365387e-10f;
}

void split_bits(int bits, int& bphi, int& by, int& bo)
{
    bo = bits / 4;
    bphi = (bits - bo + 1) / 2;
    by = bits - bo - bphi;
}

// Kernel knobs (uvrt_set_variant).  0 (default) = the top-of-tree LDS cache, leaf visits every second
// trip, refill at 8 idle lanes, 8 workgroups per CU; 400-499 = code + 10 * grid code (uvrt_extend6.hip:
// code bits 0-1 leaf period - 1, bit 2 no LDS cache; grid code 0..4 = 8 / 4 / 6 / 2 / 16 workgroups per CU)
// with refill at 16 idle lanes; 500-599 = the same with IEEE divisions everywhere; 600-899 = like 400-499
// with the refill threshold 8 / 24 / 4 idle lanes.  (The v1-v5 kernels of round 1 are gone: see git history
// and DESIGN.md section 4 for what they measured.)
bool variant_ok(int v) { return v == 0 || (v >= 400 && v < 900); }

int auto_sort_bits(int64_t n)
{
    // about one wave (64 rays) per key; no ordering for launches too small to matter
    if (n < 8192) return 0;
    int b = 0;
    while ((int64_t(64) << (b + 1)) <= n && b < 18) ++b;
    return b;
}

}  // namespace

extern "C" {

const char* uvrt_last_error(void) { return g_err.c_str(); }
const char* uvrt_version(void) { return "uvrt-mi355x 0.1 (gfx950)"; }
int uvrt_device_cus(uvrt_ctx* c) { return c ? c->num_cus : 0; }
int uvrt_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int uvrt_create(int device_id, uvrt_ctx** out)
{
    if (!out) return fail(UVRT_ERR_INVALID, "uvrt_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(UVRT_ERR_NO_DEVICE, "uvrt_create: no HIP device (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev)
        return fail(UVRT_ERR_INVALID, "uvrt_create: device %d out of range [0,%d)", device_id, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(UVRT_ERR_NO_DEVICE, "uvrt_create: device %d is %s, this library is built for gfx950 only",
                    device_id, prop.gcnArchName);
    uvrt_ctx* c = new uvrt_ctx();
    c->device = device_id;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (c->num_cus > 256) c->num_cus = 256;   // the overflow-stack buffer is sized for 256 CUs x 16 workgroups
    if (const char* e = getenv("UVRT_REPLICAS")) {   // developer knob: deposit replicas of tempPhotonMap
        const int r = atoi(e);
        if (r >= 1 && r <= 64) c->replicas_knob = r;
    }
    HIP_TRY(hipSetDevice(device_id));
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) {
        if (l > 0) HIP_TRY(hipStreamCreateWithFlags(&c->side[l], hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_tail[l], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_fence, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_mapfence, hipEventDisableTiming));
    if (const char* e = getenv("UVRT_LANES")) { const int v = atoi(e); if (v >= 1 && v <= uvrt_ctx::MAXL) c->nlanes = v; }
    if (const char* e = getenv("UVRT_PIPELINE")) c->pipeline = atoi(e) != 0;   // developer knob
    int rc = c->error_flag.ensure(256, true, c->stream);      // the flag; a developer build keeps trip statistics behind it
    // 256 CUs x 16 workgroups x 256 threads x 16 entries: the largest persistent grid
    if (!rc) rc = c->ovf_stack.ensure((size_t)OVF_MAX_ENTRIES * sizeof(uint32_t), false, c->stream);
    if (rc) { delete c; return rc; }
    *out = c;
    return UVRT_OK;
}

void uvrt_destroy(uvrt_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) {
        if (c->side[l]) (void)hipStreamSynchronize(c->side[l]);
        for (DevBuf* b : {&c->xrays[l], &c->xrecs[l], &c->xcounts[l], &c->xovf[l]}) b->release();
        if (c->ev_tail[l]) (void)hipEventDestroy(c->ev_tail[l]);
        if (c->side[l]) (void)hipStreamDestroy(c->side[l]);
    }
    if (c->ev_fence) (void)hipEventDestroy(c->ev_fence);
    if (c->ev_mapfence) (void)hipEventDestroy(c->ev_mapfence);
    if (c->comm) uvrt_comm_destroy(c);
    c->quads.release();
    for (DevBuf& b : c->recs4) b.release();
    for (DevBuf& b : c->b_recs) b.release();
    for (auto& h : c->hot) { h.perm.release(); h.hist.release(); (void)hipEventDestroy(h.ready); }
    for (auto& bset : c->bs) {
        for (DevBuf* b : {&bset.rays, &bset.planes, &bset.folded}) b->release();
        if (bset.free_ev) (void)hipEventDestroy(bset.free_ev);
    }
    for (DevBuf* b : {&c->pairs, &c->recs, &c->perm, &c->ltris, &c->leaf_count, &c->area, &c->photon_map, &c->max_map,
                      &c->counts, &c->dosage, &c->color, &c->rays, &c->keyrank, &c->sorted,
                      &c->order, &c->hits, &c->hist, &c->bin_start, &c->export_buf,
                      &c->ovf_stack, &c->error_flag})
        b->release();
    for (auto& ev : c->ev_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int uvrt_set_stream(uvrt_ctx* c, void* hip_stream)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    c->lane = 0;
    return UVRT_OK;
}

int uvrt_set_scene(uvrt_ctx* c, const void* tris64, int32_t T, const void* nodes32,
                   int32_t node_count, const uint32_t* tri_idx)
{
    if (!c || !tris64 || !nodes32 || !tri_idx) return fail(UVRT_ERR_INVALID, "uvrt_set_scene: null argument");
    if (T <= 0 || T > MAX_TRIS || node_count <= 0)
        return fail(UVRT_ERR_INVALID, "uvrt_set_scene: tri_count %d / node_count %d out of range", T, node_count);
    if (int rc = set_device(c)) return rc;

    struct HostNode { float mn[3]; int32_t leftFirst; float mx[3]; int32_t triCount; };
    static_assert(sizeof(HostNode) == 32, "BVHNode is 32 bytes (bvh.h:11-21)");
    const HostNode* nodes = (const HostNode*)nodes32;

    for (int32_t i = 0; i < T; ++i)
        if (tri_idx[i] >= (uint32_t)T)
            return fail(UVRT_ERR_BVH, "uvrt_set_scene: triIdx[%d] = %u >= tri_count %d", i, tri_idx[i], T);
    // vertex coordinates beyond 1e9 (or non-finite): the shortcuts' range proofs do not cover them
    bool huge_vertex = false;
    {
        const float* tv = (const float*)tris64;
        for (int64_t i = 0; i < (int64_t)T && !huge_vertex; ++i)
            for (int k = 0; k < 12; ++k)
                if ((k & 3) != 3 && !(std::fabs(tv[i * 16 + k]) <= 1e9f)) { huge_vertex = true; break; }
    }

    // Re-layout: walk the tree breadth-first from node 0; every inner node becomes one pair
    // record (numbered in visit order, so the top of the tree is contiguous at the front).
    auto leaf_ref = [&](const HostNode& n, std::vector<uint32_t>& leaf_count, int& err) -> uint32_t {
        int64_t first = n.leftFirst, cnt = n.triCount;
        if (first < 0 || first + cnt > T) { err = 1; return REF_DONE; }
        leaf_count[first] = (uint32_t)cnt;
        uint32_t code = cnt >= 15 ? 15u : (uint32_t)cnt;
        return REF_LEAF_BIT | (code << REF_COUNT_SHIFT) | (uint32_t)first;
    };
    std::vector<uint32_t> leaf_count(T, 0u);
    std::vector<PairRec> pairs;
    std::vector<int32_t> queue;   // inner nodes in BFS order; index in queue == pair index
    std::vector<int32_t> depth;   // tree depth of queue[i]
    uint32_t top_pairs = 0;
    int err = 0;
    bool tiny_bound = false;
    uint32_t root_ref;
    if (nodes[0].triCount > 0) {
        root_ref = leaf_ref(nodes[0], leaf_count, err);
    } else {
        root_ref = 0;
        queue.push_back(0);
        depth.push_back(0);
    }
    for (size_t qi = 0; qi < queue.size() && !err; ++qi) {
        const HostNode& n = nodes[queue[qi]];
        const int64_t l = n.leftFirst;
        if (l < 0 || l + 1 >= node_count) { err = 2; break; }
        if (queue.size() > (size_t)node_count) { err = 3; break; }   // cycle
        uint32_t ref[2];
        for (int k = 0; k < 2; ++k) {
            const HostNode& ch = nodes[l + k];
            if (ch.triCount > 0) ref[k] = leaf_ref(ch, leaf_count, err);
            else { ref[k] = (uint32_t)queue.size(); queue.push_back((int32_t)(l + k)); depth.push_back(depth[qi] + 1); }
        }
        const HostNode& a = nodes[l];
        const HostNode& b = nodes[l + 1];
        for (const HostNode* hn : {&a, &b})
            for (int k = 0; k < 3; ++k)
                for (float v : {hn->mn[k], hn->mx[k]})
                    if ((v != 0.0f && std::fabs(v) < 8.6736174e-19f) || !(std::fabs(v) <= 1e9f))
                        tiny_bound = true;   // below 2^-60, above 1e9, inf or NaN
        PairRec pr;
        pr.c0min_ref0 = make_float4(a.mn[0], a.mn[1], a.mn[2], 0.f);
        pr.c0max_ref1 = make_float4(a.mx[0], a.mx[1], a.mx[2], 0.f);
        memcpy(&pr.c0min_ref0.w, &ref[0], 4);
        memcpy(&pr.c0max_ref1.w, &ref[1], 4);
        pr.c1min = make_float4(b.mn[0], b.mn[1], b.mn[2], 0.f);
        pr.c1max = make_float4(b.mx[0], b.mx[1], b.mx[2], 0.f);
        pairs.push_back(pr);
        if (depth[qi] < 8 && top_pairs < TOP6_MAX) top_pairs = (uint32_t)qi + 1;   // level order: a prefix
    }
    if (err) return fail(UVRT_ERR_BVH, "uvrt_set_scene: malformed BVH (code %d)", err);
    // The 4-wide collapse (one level): a node takes the children of its inner children.  Numbered breadth-first
    // like the pairs; a child reference is a 64-byte unit index (2 x node index) or the BVH2 leaf reference.
    std::vector<QuadRec> quads;
    uint32_t top_quads = 0;
    if (!pairs.empty()) {
        std::vector<int32_t> qpair{0};        // pair index (of the BVH2 inner node) of every 4-wide node
        std::vector<int32_t> qdepth{0};
        const float far_box = 1e30f;          // an empty slot: a box no ray reaches (entry distance >= 1e30)
        for (size_t qi = 0; qi < qpair.size(); ++qi) {
            struct Child { float mn[3], mx[3]; uint32_t ref; };
            Child ch[4];
            int nch = 0;
            auto add = [&](const float4& mn, const float4& mx, uint32_t ref) {
                Child& c4 = ch[nch++];
                c4.mn[0] = mn.x; c4.mn[1] = mn.y; c4.mn[2] = mn.z;
                c4.mx[0] = mx.x; c4.mx[1] = mx.y; c4.mx[2] = mx.z;
                c4.ref = ref;
            };
            auto children_of = [&](const PairRec& pr, float4 mn[2], float4 mx[2], uint32_t ref[2]) {
                mn[0] = pr.c0min_ref0; mx[0] = pr.c0max_ref1; mn[1] = pr.c1min; mx[1] = pr.c1max;
                memcpy(&ref[0], &pr.c0min_ref0.w, 4);
                memcpy(&ref[1], &pr.c0max_ref1.w, 4);
            };
            float4 mn[2], mx[2];
            uint32_t ref[2];
            children_of(pairs[qpair[qi]], mn, mx, ref);
            for (int k = 0; k < 2; ++k) {
                if (ref[k] >= REF_LEAF_BIT) { add(mn[k], mx[k], ref[k]); continue; }
                float4 gmn[2], gmx[2];
                uint32_t gref[2];
                children_of(pairs[ref[k]], gmn, gmx, gref);
                for (int j = 0; j < 2; ++j) add(gmn[j], gmx[j], gref[j]);
            }
            QuadRec q;
            memset(&q, 0, sizeof q);
            float y[4][2];
            for (int k = 0; k < 4; ++k) {
                if (k < nch) {
                    uint32_t r = ch[k].ref;
                    if (r < REF_LEAF_BIT) {           // inner: it becomes a 4-wide node of its own
                        qpair.push_back((int32_t)r);
                        qdepth.push_back(qdepth[qi] + 1);
                        r = 2u * (uint32_t)(qpair.size() - 1);
                    }
                    q.xz[k] = make_float4(ch[k].mn[0], ch[k].mx[0], ch[k].mn[2], ch[k].mx[2]);
                    y[k][0] = ch[k].mn[1]; y[k][1] = ch[k].mx[1];
                    q.ref[k] = r;
                } else {
                    q.xz[k] = make_float4(far_box, far_box, far_box, far_box);
                    y[k][0] = y[k][1] = far_box;
                    q.ref[k] = REF_DONE;
                }
            }
            q.y01 = make_float4(y[0][0], y[0][1], y[1][0], y[1][1]);
            q.y23 = make_float4(y[2][0], y[2][1], y[3][0], y[3][1]);
            quads.push_back(q);
            if (qdepth[qi] < 4 && top_quads < 64) top_quads = (uint32_t)qi + 1;     // levels 0-3: up to 85 nodes, 64 cached
        }
    }
    // a child reference is a record index with 32-bit byte offsets: inner-node records + leaf records < 2^26
    if (pairs.size() + (size_t)T >= ((size_t)1 << 26))
        return fail(UVRT_ERR_INVALID, "uvrt_set_scene: %zu inner nodes + %d triangles exceed 2^26 records", pairs.size(), T);

    if (int rcj = join_all(c)) return rcj;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->lane = 0;
    const bool resized = (T != c->T);
    int rc;
    if ((rc = c->pairs.ensure(std::max<size_t>(pairs.size(), 1) * sizeof(PairRec), false, c->stream))) return rc;
    if ((rc = c->recs.ensure((pairs.size() + (size_t)T + 1) * 64, true, c->stream))) return rc;
    for (int l = 1; l < c->nlanes; ++l)
        if ((rc = c->xrecs[l].ensure((pairs.size() + (size_t)T + 1) * 64, true, c->stream))) return rc;
    // + 16 bytes: the merged record fetch of the traversal reads 64 bytes at every leaf record
    if ((rc = c->ltris.ensure((size_t)T * sizeof(LeafTri) + 16, true, c->stream))) return rc;
    if ((rc = c->leaf_count.ensure((size_t)T * 4, false, c->stream))) return rc;
    if ((rc = c->area.ensure((size_t)T * 4, false, c->stream))) return rc;
    if (resized || !c->photon_map.p) {
        // raytracer.cpp:32-37 (the reference leaves them uninitialised until reset; zero here)
        // The colour buffer stands for the GL vertex buffer of the caller's mesh (raytracer.cpp:37): a scene swap
        // does not touch it.  It only grows, so CalibratePower's detour over a 2-triangle scene
        // (raytracer.cpp:166-224, ClearBuffers(false)) leaves the room's colours as they were.
        for (DevBuf* b : {&c->photon_map, &c->max_map, &c->counts, &c->xcounts[1], &c->xcounts[2], &c->xcounts[3],
                          &c->dosage}) b->release();
        if ((rc = c->photon_map.ensure((size_t)T * 8, true, c->stream))) return rc;
        if ((rc = c->max_map.ensure((size_t)T * 8, true, c->stream))) return rc;
        // 16 deposit replicas (two per XCD: a workgroup deposits into replica blockIdx % 16), at most 64 MiB in
        // total.  Fewer than 8 serialise on the hot triangles' counters; more than ~24 push the planes out of L2 and
        // every wave then waits for its deposits' memory round trips (profiles/r02_experiments.txt: 64 replicas cost
        // the batched step 5 %)
        int R = c->replicas_knob > 0 ? c->replicas_knob : 16;
        while (R > 1 && (size_t)R * (size_t)T * 4 > ((size_t)64 << 20)) R >>= 1;
        c->replicas = R;
        if ((rc = c->counts.ensure((size_t)R * (size_t)T * 4, true, c->stream))) return rc;
        for (int l = 1; l < c->nlanes; ++l)
            if ((rc = c->xcounts[l].ensure((size_t)R * (size_t)T * 4, true, c->stream))) return rc;
        if ((rc = c->dosage.ensure((size_t)T * 4, true, c->stream))) return rc;
        if ((rc = c->color.ensure((size_t)T * 36, true, c->stream))) return rc;
    }
    if (!pairs.empty())
        HIP_TRY(hipMemcpyAsync(c->pairs.p, pairs.data(), pairs.size() * sizeof(PairRec), hipMemcpyHostToDevice, c->stream));
    if ((rc = c->quads.ensure(std::max<size_t>(quads.size(), 1) * sizeof(QuadRec), false, c->stream))) return rc;
    if (!quads.empty())
        HIP_TRY(hipMemcpyAsync(c->quads.p, quads.data(), quads.size() * sizeof(QuadRec), hipMemcpyHostToDevice, c->stream));
    for (DevBuf& b : c->recs4) b.release();              // sized per scene; rebuilt on demand (uvrt_set_wide_bvh)
    HIP_TRY(hipMemcpyAsync(c->leaf_count.p, leaf_count.data(), (size_t)T * 4, hipMemcpyHostToDevice, c->stream));
    // staging copies of the reference-layout arrays for the device-side preparation kernel
    DevBuf d_tris, d_idx;
    if ((rc = d_tris.ensure((size_t)T * 64, false, c->stream))) return rc;
    if ((rc = d_idx.ensure((size_t)T * 4, false, c->stream))) { d_tris.release(); return rc; }
    hipError_t e1 = hipMemcpyAsync(d_tris.p, tris64, (size_t)T * 64, hipMemcpyHostToDevice, c->stream);
    hipError_t e2 = hipMemcpyAsync(d_idx.p, tri_idx, (size_t)T * 4, hipMemcpyHostToDevice, c->stream);
    if (e1 == hipSuccess && e2 == hipSuccess) {
        launch_prepare_scene(d_tris.as<float4>(), d_idx.as<uint32_t>(), c->ltris.as<LeafTri>(),
                             c->area.as<float>(), T, c->stream);
        launch_prepare_leaves6(c->ltris.as<LeafTri>(), c->recs.p, (int32_t)pairs.size(), T, c->stream);
        for (int l = 1; l < c->nlanes; ++l)
            launch_prepare_leaves6(c->ltris.as<LeafTri>(), c->xrecs[l].p, (int32_t)pairs.size(), T, c->stream);
        e1 = hipGetLastError();
        e2 = hipStreamSynchronize(c->stream);
    }
    d_tris.release();
    d_idx.release();
    if (e1 != hipSuccess) return fail(UVRT_ERR_HIP, "uvrt_set_scene: %s", hipGetErrorString(e1));
    if (e2 != hipSuccess) return fail(UVRT_ERR_HIP, "uvrt_set_scene: %s", hipGetErrorString(e2));
    c->T = T;
    c->root_ref = root_ref;
    c->top_pairs = top_pairs;
    c->npairs = (int32_t)pairs.size();
    c->nquads = (int32_t)quads.size();
    c->top_quads = top_quads;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) c->recs4_valid[l] = false;
    c->recs_valid = false;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) c->xrecs_valid[l] = false;
    c->have_perm = false;
    c->have_scene = true;
    c->scene_force_exact = tiny_bound || huge_vertex;
    for (auto& h : c->hot) { h.perm.release(); h.hist.release(); (void)hipEventDestroy(h.ready); }     // statistics of the previous scene
    c->hot.clear();
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) c->lane_perm[l] = nullptr;
    // a batch of the previous scene is void; its buffers are sized per scene
    c->b_count = 0;
    c->b_is_folded = false;
    for (DevBuf& b : c->b_recs) b.release();
    c->b_recs.clear();
    c->b_recs_key.clear();
    for (auto& bset : c->bs) for (DevBuf* b : {&bset.planes, &bset.folded}) b->release();
    return UVRT_OK;
}

int uvrt_resize_rays(uvrt_ctx* c, int64_t photon_count)
{
    if (!c || photon_count < 0) return fail(UVRT_ERR_INVALID, "uvrt_resize_rays: bad argument");
    if (int rc = set_device(c)) return rc;
    if (photon_count == c->capacity && c->rays.p && (!c->record_hits || c->hits.p)) { c->last_n = -1; return UVRT_OK; }
    if (int rcj = join_all(c)) return rcj;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->lane = 0;
    int rc;
    const size_t n = (size_t)photon_count;
    if ((rc = c->rays.ensure(n * 16, false, c->stream))) return rc;
    for (int l = 1; l < c->nlanes; ++l)
        if ((rc = c->xrays[l].ensure(n * 16, false, c->stream))) return rc;
    if ((rc = c->keyrank.ensure(n * 8, false, c->stream))) return rc;
    if ((rc = c->sorted.ensure(n * 16, false, c->stream))) return rc;
    if ((rc = c->order.ensure(n * 4, false, c->stream))) return rc;
    if (c->record_hits && (rc = c->hits.ensure(n * 8, false, c->stream))) return rc;
    c->capacity = photon_count;
    c->last_n = -1;
    return UVRT_OK;
}

int uvrt_reset(uvrt_ctx* c, int32_t reset_color)
{
    if (!c || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_reset: no scene");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    launch_reset(c->photon_map.as<double>(), c->max_map.as<double>(), c->counts.as<int32_t>(),
                 c->replicas, c->T, c->color.as<float>(), reset_color, c->T, c->stream);
    HIP_TRY(hipGetLastError());
    // side-lane count buffers are zero unless an extend was never accumulated; only then must later
    // generate / extend work on the side streams wait for this reset
    bool dirty = false;
    for (int l = 1; l < uvrt_ctx::MAXL; ++l) {
        if (c->counts_dirty[l] && c->xcounts[l].p) {
            HIP_TRY(hipMemsetAsync(c->xcounts[l].p, 0, c->xcounts[l].bytes, c->stream));
            dirty = true;
        }
        c->counts_dirty[l] = false;
    }
    c->counts_dirty[0] = false;
    if (c->b_count > 0) {     // a traced batch that was never replayed: drop its deposits
        if (c->b_is_folded) HIP_TRY(hipMemsetAsync(c->bs[c->b_set].folded.p, 0, c->bs[c->b_set].folded.bytes, c->stream));
        else HIP_TRY(hipMemsetAsync(c->bs[c->b_set].planes.p, 0, c->bs[c->b_set].planes.bytes, c->stream));
        HIP_TRY(hipEventRecord(c->bs[c->b_set].free_ev, c->stream));
        c->b_count = 0;
        c->b_is_folded = false;
    }
    return dirty ? mark_fence(c) : mark_map_fence(c);
}

uint32_t uvrt_seed_next(const float lp[3], float light_length, uint32_t seed_prev)
{
    return uvrt_seed_next_mode(lp, light_length, seed_prev, 0);
}

uint32_t uvrt_seed_next_mode(const float lp[3], float light_length, uint32_t seed_prev, int32_t seed_mode)
{
    // work-item 0 of cl/generate.cl:13-39; the ray itself is not needed, only the RNG state
    float acc = (float)(0 * 17 + 1);
    acc = acc + lp[0] * 13.0f;
    acc = acc + lp[1] * 7.0f;
    acc = acc + lp[2] * 11.0f;
    acc = acc + (float)(seed_prev >> 15);
    uint32_t seed = host_wang_hash((seed_mode == 1 && acc < 0.0f) ? 0u : (uint32_t)(int64_t)acc);
    (void)light_length;
    (void)host_random_float(seed);   // origin.y
    (void)host_random_float(seed);   // diry
    double x = (double)(host_random_float(seed) * 2.0f - 1.0f);
    double y = (double)(host_random_float(seed) * 2.0f - 1.0f);
    while (x * x + y * y > 1.0) {
        x = (double)(host_random_float(seed) * 2.0f - 1.0f);
        y = (double)(host_random_float(seed) * 2.0f - 1.0f);
    }
    return seed;
}

int uvrt_generate(uvrt_ctx* c, const float lp[3], float light_length, int64_t first_gid, int64_t n)
{
    if (!c || !lp) return fail(UVRT_ERR_INVALID, "uvrt_generate: null argument");
    if (n < 0 || first_gid < 0 || n > c->capacity)
        return fail(UVRT_ERR_INVALID, "uvrt_generate: n = %lld exceeds the ray capacity %lld (uvrt_resize_rays)",
                    (long long)n, (long long)c->capacity);
    if (first_gid + n > (int64_t)INT32_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_generate: global id beyond int range (generate.cl:11)");
    if (int rc = set_device(c)) return rc;

    const uint32_t seed_prev = c->seed;
    const uint32_t seed_next = uvrt_seed_next_mode(lp, light_length, seed_prev, c->seed_mode);

    int bits = c->sort_bits < 0 ? auto_sort_bits(n) : c->sort_bits;
    if (bits > 20) bits = 20;
    // launch lane: alternate between the two streams / buffer sets when nothing stands against it
    {
        const bool pipe_ok = c->pipeline && c->nlanes > 1 && !c->record_hits && bits == 0 &&
                             c->xrays[1].p;
        c->prev_lane = c->lane;
        c->cur_pipelined = pipe_ok;
        if (pipe_ok) c->lane = (c->lane + 1) % c->nlanes;
        else { if (int rc = join_all(c)) return rc; c->lane = 0; }
        if (c->lane != 0) {
            // 8 workgroups per CU x 256 threads x 24 overflow entries (the largest grid a side lane runs)
            if (int rc = c->xovf[c->lane].ensure((size_t)c->num_cus * 8 * 256 * 24 * sizeof(uint32_t), false, c->stream)) return rc;
        }
    }
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls)) return rc;
    GenParams p;
    memset(&p, 0, sizeof p);
    p.rays = lane_rays(c).as<float4>();
    p.lx = lp[0]; p.ly = lp[1]; p.lz = lp[2];
    p.light_length = light_length;
    p.first_gid = first_gid;
    p.n = n;
    p.seed_prev = seed_prev;
    p.seed_next = seed_next;
    p.seed_mode = c->seed_mode;
    if (bits > 0 && n > 0) {
        const int32_t nbins = 1 << bits;
        if (c->hist_bins < nbins) {
            int rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->hist.release();
            c->bin_start.release();
            if ((rc = c->hist.ensure((size_t)nbins * 4, true, c->stream))) return rc;
            if ((rc = c->bin_start.ensure((size_t)nbins * 4, true, c->stream))) return rc;
            c->hist_bins = nbins;
        }
        p.keyrank = c->keyrank.as<uint2>();
        p.hist = c->hist.as<uint32_t>();
        split_bits(bits, p.bits_phi, p.bits_y, p.bits_o);
    }
    if (c->npairs > 0) {   // extend's per-launch records ride along in the same launch
        p.prep_pairs = c->pairs.as<PairRec>();
        p.prep_recs = lane_recs(c).as<float4>();
        // the statistics always sample global ids [0, 32768) of the lamp (the kernel makes its own rays), whichever
        // range of the launch this context traces; launches too small to repay them keep the breadth-first order
        const uint32_t* pm = c->have_perm ? c->perm.as<uint32_t>() : nullptr;
        if (!pm && n >= 16384)
            if (int rc = launch_perm(c, lp, light_length, seed_prev, seed_next, 32768, ls, &pm)) return rc;
        c->lane_perm[c->lane] = pm;
        p.prep_perm = pm;
        p.prep_npairs = c->npairs;
    }
    launch_generate(p, ls);
    HIP_TRY(hipGetLastError());
    (c->lane ? c->xrecs_valid[c->lane] : c->recs_valid) = p.prep_recs != nullptr;
    (c->lane ? c->xrecs_ox[c->lane] : c->recs_ox) = lp[0];
    (c->lane ? c->xrecs_oz[c->lane] : c->recs_oz) = lp[2];
    if (p.keyrank) {
        launch_scan_bins(c->hist.as<uint32_t>(), c->bin_start.as<uint32_t>(), 1 << bits, c->stream);
        launch_scatter(c->rays.as<float4>(), c->keyrank.as<uint2>(), c->bin_start.as<uint32_t>(),
                       c->sorted.as<float4>(), c->order.as<uint32_t>(), n, c->stream);
        HIP_TRY(hipGetLastError());
    }
    c->seed = seed_next;
    c->last_n = n;
    c->last_first = first_gid;
    c->last_sorted = p.keyrank != nullptr;
    c->last_extended = false;
    c->ox = lp[0];
    c->oz = lp[2];
    return UVRT_OK;
}

int uvrt_extend(uvrt_ctx* c, int64_t n)
{
    if (!c || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_extend: no scene");
    if (c->last_n < 0 || n != c->last_n)
        return fail(UVRT_ERR_INVALID, "uvrt_extend: n = %lld does not match the last generate (%lld)",
                    (long long)n, (long long)c->last_n);
    if (int rc = set_device(c)) return rc;
    if (c->record_hits) {
        if (int rc = c->hits.ensure((size_t)c->capacity * 8, false, c->stream)) return rc;
    }
    ExtendParams p;
    memset(&p, 0, sizeof p);
    p.scene.pairs = c->pairs.as<PairRec>();
    p.scene.ltris = c->ltris.as<LeafTri>();
    p.scene.leaf_count = c->leaf_count.as<uint32_t>();
    p.scene.root_ref = c->root_ref;
    p.scene.tri_count = c->T;
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls)) return rc;
    p.rays = c->last_sorted ? c->sorted.as<float4>() : lane_rays(c).as<float4>();
    {
        // conditions of the reciprocal shortcut that are uniform over the launch (slab<>())
        const float ax = std::fabs(c->ox), az = std::fabs(c->oz);
        const float tiny = 7.888609e-31f;   // 2^-100
        p.force_exact = (c->scene_force_exact || (ax != 0.0f && ax < tiny) || (az != 0.0f && az < tiny) ||
                         !(ax <= 1e9f) || !(az <= 1e9f)) ? 1 : 0;
    }
    p.order = c->last_sorted ? c->order.as<uint32_t>() : nullptr;
    p.hits = c->record_hits ? c->hits.as<uint2>() : nullptr;
    p.ovf_stack = lane_ovf(c).as<uint32_t>();
    p.ovf_capacity = lane_ovf(c).bytes / sizeof(uint32_t);
    p.num_cus = c->num_cus;
    p.flavour = c->flavour;
    p.top_pairs = c->top_pairs;
    p.counts = lane_counts(c).as<int32_t>();
    p.count_replicas = c->replicas;
    p.count_stride = c->T;
    p.error_flag = c->error_flag.as<uint32_t>();
    p.ox = c->ox;
    p.oz = c->oz;
    p.n = n;
    p.npairs = c->npairs;
    p.recs = lane_recs(c).p;
    p.perm = c->have_perm ? c->perm.as<uint32_t>() : c->lane_perm[c->lane];
    {
        const bool valid = c->lane ? c->xrecs_valid[c->lane] : c->recs_valid;
        const float rox = c->lane ? c->xrecs_ox[c->lane] : c->recs_ox, roz = c->lane ? c->xrecs_oz[c->lane] : c->recs_oz;
        p.recs_prepared = (valid && memcmp(&rox, &c->ox, 4) == 0 && memcmp(&roz, &c->oz, 4) == 0) ? 1 : 0;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        if (c->ev_used == c->ev_pool.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            c->ev_pool.emplace_back(a, b);
        }
        e0 = c->ev_pool[c->ev_used].first;
        e1 = c->ev_pool[c->ev_used].second;
        ++c->ev_used;
        HIP_TRY(hipEventRecord(e0, ls));
    }
    if (c->wide && c->nquads > 0) {
        // the opt-in 4-wide walk: its per-launch records are (re)made here when the lane's are for another lamp
        DevBuf& r4 = c->recs4[c->lane];
        if (!r4.p) {
            if (int rc = r4.ensure(((size_t)2 * c->nquads + (size_t)c->T + 1) * 64, true, ls)) return rc;
            launch_prepare_leaves6(c->ltris.as<LeafTri>(), r4.p, 2 * c->nquads, c->T, ls);
            c->recs4_valid[c->lane] = false;
        }
        if (!c->recs4_valid[c->lane] || memcmp(&c->recs4_ox[c->lane], &c->ox, 4) != 0 || memcmp(&c->recs4_oz[c->lane], &c->oz, 4) != 0) {
            launch_prepare_launch4(c->quads.as<QuadRec>(), r4.p, c->ox, c->oz, c->nquads, ls);
            c->recs4_valid[c->lane] = true;
            c->recs4_ox[c->lane] = c->ox;
            c->recs4_oz[c->lane] = c->oz;
        }
        p.recs4 = r4.p;
        p.nquads = c->nquads;
        p.top_quads = c->top_quads;
        p.refill_min = 8;
        if (c->variant >= 500 && c->variant < 600) p.force_exact = 1;
        if (!launch_extend4(p, 7, ls)) return fail(UVRT_ERR_INVALID, "uvrt_extend: overflow-stack buffer too small for the 4-wide kernel");
        HIP_TRY(hipGetLastError());
        if (c->timing) HIP_TRY(hipEventRecord(e1, ls));
        c->counts_dirty[c->lane] = true;
        c->last_extended = c->record_hits;
        return UVRT_OK;
    }
    if (c->variant >= 500 && c->variant < 600) p.force_exact = 1;
    p.refill_min = c->variant == 0 ? 8 : c->variant >= 800 ? 4 : c->variant >= 700 ? 24 : c->variant >= 600 ? 8 : 16;
    static const int per_cu[6] = {8, 4, 6, 2, 16, 7};
    const int gcode = (c->variant / 10) % 10;
    const int code6 = c->variant == 0 ? 1 : c->variant % 10;   // default: LDS top cache, leaf visits every 2nd trip
    // default grid: 8 workgroups per CU on one stream (20 KB of LDS each: eight fit a CU); 7 when launches are
    // pipelined over several streams -- the free slot per CU lets the first workgroups of the next launch and the
    // small kernels around it (generate, accumulate, replay) run at once instead of queueing behind persistent waves
    // (profiles/r02_experiments.txt); with four launch lanes 4 per CU
    const int per_cu_default = (c->cur_pipelined && c->nlanes >= 4) ? 4 : c->cur_pipelined ? 7 : 8;
    if (!launch_extend6(p, code6, c->variant == 0 ? per_cu_default : per_cu[gcode < 6 ? gcode : 0], ls))
        return fail(UVRT_ERR_INVALID, "uvrt_extend: variant %d needs a larger overflow-stack buffer than the context holds", c->variant);
    HIP_TRY(hipGetLastError());
    if (c->timing) HIP_TRY(hipEventRecord(e1, ls));
    c->counts_dirty[c->lane] = true;
    c->last_extended = c->record_hits;
    return UVRT_OK;
}

int uvrt_accumulate(uvrt_ctx* c, float time_step, int32_t tri_count)
{
    if (!c || !c->have_scene || tri_count < 0 || tri_count > c->T)
        return fail(UVRT_ERR_INVALID, "uvrt_accumulate: bad tri_count");
    if (int rc = set_device(c)) return rc;
    // the maps are updated in launch order: wait for whatever the other lane has enqueued so far
    // (its accumulate and shade), not for this lane's successor
    if (int rc = order_after_previous(c)) return rc;
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls, true)) return rc;
    launch_accumulate(c->photon_map.as<double>(), c->max_map.as<double>(), lane_counts(c).as<int32_t>(),
                      c->replicas, c->T, time_step, tri_count, ls);
    if (tri_count == c->T) c->counts_dirty[c->lane] = false;
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}

int uvrt_compute_dosage(uvrt_ctx* c, int32_t which, int32_t photons_per_light, float scaled_power,
                        int32_t tri_count)
{
    if (!c || !c->have_scene || tri_count < 0 || tri_count > c->T)
        return fail(UVRT_ERR_INVALID, "uvrt_compute_dosage: bad tri_count");
    if (which != UVRT_MAP_SUM && which != UVRT_MAP_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_compute_dosage: which_map must be 0 or 1");
    if (int rc = set_device(c)) return rc;
    const double* map = which == UVRT_MAP_SUM ? c->photon_map.as<double>() : c->max_map.as<double>();
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls, true)) return rc;
    launch_compute_dosage(map, c->dosage.as<float>(), c->area.as<float>(), photons_per_light,
                          scaled_power, tri_count, ls);
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}

int uvrt_dosage_to_color(uvrt_ctx* c, float min_value, int32_t threshold_view, int32_t tri_count)
{
    if (!c || !c->have_scene || tri_count < 0 || tri_count > c->T)
        return fail(UVRT_ERR_INVALID, "uvrt_dosage_to_color: bad tri_count");
    if (int rc = set_device(c)) return rc;
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls, true)) return rc;
    launch_dosage_to_color(c->dosage.as<float>(), c->color.as<float>(), min_value, threshold_view,
                           tri_count, ls);
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}

int uvrt_set_record_perm(uvrt_ctx* c, const uint32_t* perm, int32_t n)
{
    if (!c || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_set_record_perm: no scene");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->recs_valid = false;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) c->xrecs_valid[l] = false;
    if (!perm) { c->have_perm = false; return UVRT_OK; }
    if (n != c->npairs) return fail(UVRT_ERR_INVALID, "uvrt_set_record_perm: %d entries, the scene has %d inner nodes", n, c->npairs);
    std::vector<uint8_t> seen((size_t)n, 0);
    for (int32_t i = 0; i < n; ++i) {
        if (perm[i] >= (uint32_t)n || seen[perm[i]]) return fail(UVRT_ERR_INVALID, "uvrt_set_record_perm: not a permutation");
        seen[perm[i]] = 1;
    }
    if (n == 0) { c->have_perm = false; return UVRT_OK; }
    if (int rc = c->perm.ensure((size_t)n * 4, false, c->stream)) return rc;
    HIP_TRY(hipMemcpyAsync(c->perm.p, perm, (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_perm = true;
    return UVRT_OK;
}

int uvrt_shade(uvrt_ctx* c, int32_t which, int32_t photons_per_light, float scaled_power, float min_value,
               int32_t threshold_view, int32_t tri_count)
{
    if (!c || !c->have_scene || tri_count < 0 || tri_count > c->T)
        return fail(UVRT_ERR_INVALID, "uvrt_shade: bad tri_count");
    if (which != UVRT_MAP_SUM && which != UVRT_MAP_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_shade: which_map must be 0 or 1");
    if (int rc = set_device(c)) return rc;
    const double* map = which == UVRT_MAP_SUM ? c->photon_map.as<double>() : c->max_map.as<double>();
    hipStream_t ls;
    if (int rc = lane_stream(c, &ls, true)) return rc;
    launch_shade(map, c->dosage.as<float>(), c->area.as<float>(), c->color.as<float>(), photons_per_light,
                 scaled_power, min_value, threshold_view, tri_count, ls);
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}

int uvrt_sync(uvrt_ctx* c)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    uint32_t flag = 0;
    HIP_TRY(hipMemcpyAsync(&flag, c->error_flag.p, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
#ifdef UVRT_TRIP_STATS
    if (getenv("UVRT_TRIP_STATS")) {
        unsigned long long st[21];
        HIP_TRY(hipMemcpy(st, (char*)c->error_flag.p + 8, sizeof st, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemset((char*)c->error_flag.p + 8, 0, sizeof st));
        if (st[1]) {
            const double t = (double)st[1];
            fprintf(stderr, "trip stats: waves %llu trips %llu (%.1f per wave)  lanes per trip: inner %.2f leaf-visit %.2f leaf-wait %.2f "
                    "idle %.2f  cached %.2f  leaf trips %.3f  drain trips %.3f  slow trips %.4f (sp>=9 %.4f >=10 %.4f >=11 %.4f >=12 %.4f)  "
                    "refills/wave %.1f\n",
                    st[0], st[1], t / (double)st[0], st[2] / t, st[3] / t, st[4] / t, st[5] / t, st[10] / t, st[6] / t, st[7] / t,
                    st[8] / t, st[11] / t, st[12] / t, st[13] / t, st[14] / t, (double)st[9] / (double)st[0]);
            const double w = (double)st[0];
            fprintf(stderr, "trip clocks (s_memtime ticks per wave): life %.0f  record fetch %.0f  leaf tests %.0f  box tests + descend %.0f  "
                    "refill %.0f  general step %.0f  rest (loop overhead) %.0f;  per trip: fetch %.0f leaf %.0f box %.0f; per refill %.0f; per general step %.0f\n",
                    st[19] / w, st[15] / w, st[16] / w, st[17] / w, st[18] / w, st[20] / w,
                    (st[19] - st[15] - st[16] - st[17] - st[18] - st[20]) / w,
                    st[15] / t, st[16] / t, st[17] / t, (double)st[18] / (double)(st[9] ? st[9] : 1), (double)st[20] / (double)(st[8] ? st[8] : 1));
        }
    }
#endif
    if (flag) {
        HIP_TRY(hipMemsetAsync(c->error_flag.p, 0, 4, c->stream));
        return fail(UVRT_ERR_STACK, "extend: BVH traversal needed more than 32 stack entries (extend.cl:43)");
    }
    return UVRT_OK;
}

static int read_back(uvrt_ctx* c, const DevBuf& b, size_t elem, void* out, int64_t first, int64_t count,
                     int64_t limit, const char* what)
{
    if (!c || !out || first < 0 || count < 0 || first + count > limit)
        return fail(UVRT_ERR_INVALID, "%s: range [%lld,+%lld) outside [0,%lld)", what, (long long)first,
                    (long long)count, (long long)limit);
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    if (count == 0) return UVRT_OK;
    HIP_TRY(hipMemcpyAsync(out, (const char*)b.p + (size_t)first * elem, (size_t)count * elem,
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UVRT_OK;
}

int uvrt_read_dosage(uvrt_ctx* c, float* out, int32_t first, int32_t count)
{
    return read_back(c, c ? c->dosage : DevBuf(), 4, out, first, count, c ? c->T : 0, "uvrt_read_dosage");
}
int uvrt_read_color(uvrt_ctx* c, float* out9, int32_t first, int32_t count)
{
    return read_back(c, c ? c->color : DevBuf(), 36, out9, first, count, c ? c->T : 0, "uvrt_read_color");
}
int uvrt_read_counts(uvrt_ctx* c, int32_t* out, int32_t first, int32_t count)
{
    if (c && c->have_scene) {
        if (int rc = set_device(c)) return rc;
        if (int rc = join_all(c)) return rc;
        launch_fold_counts(lane_counts(c).as<int32_t>(), c->replicas, c->T, c->T, c->stream);
        if (int rc = mark_fence(c)) return rc;
    }
    return read_back(c, c ? lane_counts(c) : DevBuf(), 4, out, first, count, c ? c->T : 0, "uvrt_read_counts");
}
int uvrt_read_photon_map(uvrt_ctx* c, int32_t which, double* out, int32_t first, int32_t count)
{
    if (which != UVRT_MAP_SUM && which != UVRT_MAP_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_read_photon_map: which_map must be 0 or 1");
    return read_back(c, c ? (which == UVRT_MAP_SUM ? c->photon_map : c->max_map) : DevBuf(), 8, out, first,
                     count, c ? c->T : 0, "uvrt_read_photon_map");
}

int uvrt_get_seed(uvrt_ctx* c, uint32_t* seed)
{
    if (!c || !seed) return fail(UVRT_ERR_INVALID, "null argument");
    *seed = c->seed;
    return UVRT_OK;
}
int uvrt_set_seed(uvrt_ctx* c, uint32_t seed)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    c->seed = seed;
    return UVRT_OK;
}


// ---------------------------------------------------------------- batched tracing, include/uvrt.h

int uvrt_trace_batch(uvrt_ctx* c, const float* lamps, float light_length, int32_t count, int64_t first_gid, int64_t n)
{
    if (!c || !lamps || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: null argument or no scene");
    if (count <= 0 || count > MAX_BATCH) return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: count must be in [1,%d]", MAX_BATCH);
    if (n <= 0 || first_gid < 0 || first_gid + n > (int64_t)INT32_MAX)
        return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: bad global-id range");
    if (c->b_count > 0) return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: the previous batch has not been replayed (uvrt_replay_batch)");
    if (c->record_hits || c->sort_bits != 0)
        return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: per-ray hit records and ray ordering are per-launch features");
    if (int rc = set_device(c)) return rc;
    const int64_t n_pad = (n + 63) / 64 * 64;
    // deposit replicas per plane: the contention on a hot triangle's counter grows with the rays per plane
    // (16 replicas for 2 M rays), and every replica is read and zeroed again by the replay -- a shard of a launch
    // gets by with 8 (one per XCD)
    int R = c->replicas;
    while (R > 8 && (int64_t)R * 131072 > 2 * n) R >>= 1;
    if ((uint64_t)count * (uint64_t)n_pad >= ((uint64_t)1 << 30) || (uint64_t)count * (uint64_t)R * (uint64_t)c->T >= ((uint64_t)1 << 32))
        return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: %d launches x %lld rays exceed one batch (2^30 ray slots, 2^32 counters)", count, (long long)n);

    // group the launches by lamp column (x, z): the per-launch node-pair records depend on it only
    int group_of[MAX_BATCH], ngroups = 0, gfirst[MAX_BATCH], gsize[MAX_BATCH] = {};
    float gx[MAX_BATCH], gz[MAX_BATCH];
    for (int k = 0; k < count; ++k) {
        int g = 0;
        for (; g < ngroups; ++g)
            if (memcmp(&gx[g], &lamps[3 * k], 4) == 0 && memcmp(&gz[g], &lamps[3 * k + 2], 4) == 0) break;
        if (g == ngroups) { gx[g] = lamps[3 * k]; gz[g] = lamps[3 * k + 2]; ++ngroups; }
        group_of[k] = g;
        ++gsize[g];
    }
    for (int g = 0, acc = 0; g < ngroups; ++g) { gfirst[g] = acc; acc += gsize[g]; }
    GenBatchParams gp;
    memset(&gp, 0, sizeof gp);
    {
        int fill[MAX_BATCH] = {};
        uint32_t seed = c->seed;
        for (int k = 0; k < count; ++k) {                    // logical order: the SEED chain
            const int g = group_of[k], ph = gfirst[g] + fill[g]++;
            c->b_phys[k] = ph;
            gp.lx[ph] = lamps[3 * k]; gp.ly[ph] = lamps[3 * k + 1]; gp.lz[ph] = lamps[3 * k + 2];
            gp.seed_prev[ph] = seed;
            seed = uvrt_seed_next_mode(&lamps[3 * k], light_length, seed, c->seed_mode);
            gp.seed_next[ph] = seed;
        }
        c->seed = seed;
    }
    // The batch goes into the buffer set the previous batch did NOT use: its lanes start at once -- in the drain of
    // the previous batch, while that one is still being folded / reduced / replayed on the context's stream -- and
    // only wait for the set's last replay (free_ev), which is two batches back.  Anything that has to touch memory
    // the lanes may still read (growing a buffer, new per-launch records) first waits for everything.
    const int set = c->b_set ^ 1;
    uvrt_ctx::BatchSet& S = c->bs[set];
    int rc;
    const size_t plane_ints = (size_t)R * (size_t)c->T;
    // full planes are allocated for the context's replica count: R only shrinks the part of it that is used
    const size_t plane_alloc = (size_t)c->replicas * (size_t)c->T;
    bool need_sync = S.rays.bytes < (size_t)count * (size_t)n_pad * 16 || S.planes.bytes < (size_t)count * plane_alloc * 4 ||
                     S.folded.bytes < (size_t)count * (size_t)c->T * 4 || (int)c->b_recs.size() < ngroups || !S.free_ev;
    const uint32_t* gperm[MAX_BATCH] = {};
    for (int g = 0; g < ngroups; ++g) {
        gperm[g] = c->have_perm ? c->perm.as<uint32_t>() : nullptr;
        if (!gperm[g] && (int64_t)gsize[g] * n >= 16384) {
            const int ph = gfirst[g];      // the group's first launch lends its lamp and seeds to the statistics
            const float gl[3] = {gp.lx[ph], gp.ly[ph], gp.lz[ph]};
            if (int rcp = launch_perm(c, gl, light_length, gp.seed_prev[ph], gp.seed_next[ph], 32768, c->stream, &gperm[g])) return rcp;
        }
        if (g >= (int)c->b_recs_key.size() || c->b_recs_key[g].perm != gperm[g] || memcmp(&c->b_recs_key[g].ox, &gx[g], 4) != 0 ||
            memcmp(&c->b_recs_key[g].oz, &gz[g], 4) != 0)
            need_sync = true;
    }
    if (need_sync) {
        if (int rcj = join_all(c)) return rcj;
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (!S.free_ev) HIP_TRY(hipEventCreateWithFlags(&S.free_ev, hipEventDisableTiming));
        const bool grown = S.planes.bytes < (size_t)count * plane_alloc * 4 || S.folded.bytes < (size_t)count * (size_t)c->T * 4;
        if ((rc = S.rays.ensure((size_t)count * (size_t)n_pad * 16, false, c->stream))) return rc;
        if ((rc = S.planes.ensure((size_t)count * plane_alloc * 4, true, c->stream))) return rc;
        if ((rc = S.folded.ensure((size_t)count * (size_t)c->T * 4, true, c->stream))) return rc;
        if (grown) HIP_TRY(hipEventRecord(S.free_ev, c->stream));      // the zero fill is the set's "last replay"
        while ((int)c->b_recs.size() < ngroups) {
            DevBuf b;
            if ((rc = b.ensure(((size_t)c->npairs + (size_t)c->T + 1) * 64, true, c->stream))) return rc;
            launch_prepare_leaves6(c->ltris.as<LeafTri>(), b.p, c->npairs, c->T, c->stream);
            c->b_recs.push_back(b);
        }
        c->b_recs_key.resize(c->b_recs.size());
        for (int l = 1; l <= 2; ++l)
            if ((rc = c->xovf[l].ensure((size_t)c->num_cus * 8 * 256 * 24 * sizeof(uint32_t), false, c->stream))) return rc;
        // per-launch records of the lamp columns whose array holds something else
        for (int g = 0; g < ngroups; ++g) {
            uvrt_ctx::RecsKey& key = c->b_recs_key[g];
            if (key.perm == gperm[g] && key.valid && memcmp(&key.ox, &gx[g], 4) == 0 && memcmp(&key.oz, &gz[g], 4) == 0) continue;
            launch_prepare_launch6(c->pairs.as<PairRec>(), c->b_recs[g].p, gx[g], gz[g], c->npairs, gperm[g], c->stream);
            key.ox = gx[g]; key.oz = gz[g]; key.perm = gperm[g]; key.valid = true;
        }
        HIP_TRY(hipGetLastError());
        if (int rcf = mark_fence(c)) return rcf;         // the lanes' next work waits for the records
    }
    // Launches in CHUNKS of a few planes: generate + fused extend of a chunk on one launch lane, chunks alternating
    // over the lanes.  A chunk's rays (16 B each) are sized to stay in the Infinity Cache between the generate
    // that writes them and the extend that reads them (a refill that has to go to HBM stalls its wave for
    // microseconds), and the next chunk's generate and first waves run in the drain of the previous one.
    bool lane_waited[uvrt_ctx::MAXL] = {};
    size_t chunk_bytes = (size_t)96 << 20;
    if (const char* e = getenv("UVRT_BATCH_CHUNK_MB")) { const long v = atol(e); if (v > 0) chunk_bytes = (size_t)v << 20; }
    const int per_chunk = (int)std::max<size_t>(1, chunk_bytes / ((size_t)n_pad * 16));
    const int lane_before = c->lane;
    int chunk_index = 0;
    for (int g = 0; g < ngroups; ++g) {
        for (int k0 = 0; k0 < gsize[g]; k0 += per_chunk, ++chunk_index) {
            const int kc = std::min(per_chunk, gsize[g] - k0), ph0 = gfirst[g] + k0;
            // two SIDE lanes in turn: the context's own stream carries the fold / reduce / replay of the previous batch,
            // which a chunk enqueued there would have to wait for
            c->lane = c->pipeline ? 1 + (int)(c->b_chunks++ & 1u) : 0;
            hipStream_t ls;
            if (int rcl = lane_stream(c, &ls)) { c->lane = lane_before; return rcl; }
            if (!lane_waited[c->lane]) {      // the set's previous occupant has been replayed (two batches back)
                HIP_TRY(hipStreamWaitEvent(ls, S.free_ev, 0));
                lane_waited[c->lane] = true;
            }
            GenBatchParams gq;
            memset(&gq, 0, sizeof gq);
            gq.rays = S.rays.as<float4>() + (size_t)ph0 * (size_t)n_pad;
            gq.n_pad = n_pad;
            gq.first_gid = first_gid;
            gq.n = n;
            gq.light_length = light_length;
            gq.seed_mode = c->seed_mode;
            gq.count = kc;
            for (int j = 0; j < kc; ++j) {
                gq.lx[j] = gp.lx[ph0 + j]; gq.ly[j] = gp.ly[ph0 + j]; gq.lz[j] = gp.lz[ph0 + j];
                gq.seed_prev[j] = gp.seed_prev[ph0 + j]; gq.seed_next[j] = gp.seed_next[ph0 + j];
            }
            launch_generate_batch(gq, ls);
            ExtendParams p;
            memset(&p, 0, sizeof p);
            p.scene.pairs = c->pairs.as<PairRec>();
            p.scene.ltris = c->ltris.as<LeafTri>();
            p.scene.leaf_count = c->leaf_count.as<uint32_t>();
            p.scene.root_ref = c->root_ref;
            p.scene.tri_count = c->T;
            p.rays = gq.rays;
            {
                const float ax = std::fabs(gx[g]), az = std::fabs(gz[g]);
                const float tiny = 7.888609e-31f;   // 2^-100
                p.force_exact = (c->scene_force_exact || (ax != 0.0f && ax < tiny) || (az != 0.0f && az < tiny) ||
                                 !(ax <= 1e9f) || !(az <= 1e9f) || (c->variant >= 500 && c->variant < 600)) ? 1 : 0;
            }
            p.ovf_stack = lane_ovf(c).as<uint32_t>();
            p.ovf_capacity = lane_ovf(c).bytes / sizeof(uint32_t);
            p.num_cus = c->num_cus;
            p.flavour = c->flavour;
            p.top_pairs = c->top_pairs;
            p.counts = S.planes.as<int32_t>() + (size_t)ph0 * plane_ints;
            p.count_replicas = R;
            p.count_stride = c->T;
            p.error_flag = c->error_flag.as<uint32_t>();
            p.ox = gx[g];
            p.oz = gz[g];
            p.n = (int64_t)kc * n_pad;
            p.npairs = c->npairs;
            p.recs = c->b_recs[g].p;
            p.perm = gperm[g];
            p.recs_prepared = 1;
            p.refill_min = c->variant == 0 ? 8 : c->variant >= 800 ? 4 : c->variant >= 700 ? 24 : c->variant >= 600 ? 8 : 16;
            p.plane_batches = (uint32_t)(n_pad / 64);
            p.plane_n = (uint32_t)n;
            p.plane_stride = (uint32_t)plane_ints;
            static const int per_cu[6] = {8, 4, 6, 2, 16, 7};
            const int gcode = (c->variant / 10) % 10;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (c->timing) {
                if (c->ev_used == c->ev_pool.size()) {
                    hipEvent_t a, b;
                    HIP_TRY(hipEventCreate(&a));
                    HIP_TRY(hipEventCreate(&b));
                    c->ev_pool.emplace_back(a, b);
                }
                e0 = c->ev_pool[c->ev_used].first;
                e1 = c->ev_pool[c->ev_used].second;
                ++c->ev_used;
                HIP_TRY(hipEventRecord(e0, ls));
            }
            if (!launch_extend6(p, c->variant == 0 ? 1 : c->variant % 10, c->variant == 0 ? (c->pipeline ? 7 : 8) : per_cu[gcode < 6 ? gcode : 0], ls)) {
                c->lane = lane_before;
                return fail(UVRT_ERR_INVALID, "uvrt_trace_batch: variant %d needs a larger overflow-stack buffer", c->variant);
            }
            HIP_TRY(hipGetLastError());
            if (c->timing) HIP_TRY(hipEventRecord(e1, ls));
        }
    }
    c->lane = 0;
    c->cur_pipelined = false;
    c->last_n = -1;                      // the per-launch generate/extend pairing starts afresh
    c->b_set = set;
    c->b_repl = R;
    c->b_count = count;
    c->b_n = n;
    c->b_npad = n_pad;
    c->b_is_folded = false;
    return UVRT_OK;
}

int uvrt_fold_batch(uvrt_ctx* c)
{
    if (!c || c->b_count <= 0) return fail(UVRT_ERR_INVALID, "uvrt_fold_batch: no traced batch");
    if (c->b_is_folded) return UVRT_OK;
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    launch_fold_planes(c->bs[c->b_set].planes.as<int32_t>(), c->bs[c->b_set].folded.as<int32_t>(), c->b_count, c->b_repl, c->T, c->stream);
    HIP_TRY(hipGetLastError());
    c->b_is_folded = true;
    return UVRT_OK;          // on the context's stream like everything else that touches the set until its replay
}

int uvrt_replay_batch(uvrt_ctx* c, const uvrt_replay_op* ops, int32_t count, int32_t tri_count)
{
    if (!c || !ops || c->b_count <= 0) return fail(UVRT_ERR_INVALID, "uvrt_replay_batch: no traced batch");
    if (count != c->b_count) return fail(UVRT_ERR_INVALID, "uvrt_replay_batch: %d operations for a batch of %d launches", count, c->b_count);
    if (tri_count < 0 || tri_count > c->T) return fail(UVRT_ERR_INVALID, "uvrt_replay_batch: bad tri_count");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    ReplayParams p;
    memset(&p, 0, sizeof p);
    p.photon_map = c->photon_map.as<double>();
    p.max_map = c->max_map.as<double>();
    p.planes = c->bs[c->b_set].planes.as<int32_t>();
    p.folded = c->bs[c->b_set].folded.as<int32_t>();
    p.dosage = c->dosage.as<float>();
    p.color = c->color.as<float>();
    p.area = c->area.as<float>();
    p.plane_stride = (int64_t)c->b_repl * c->T;
    p.replicas = c->b_repl;
    p.T = tri_count;
    p.count = count;
    p.is_folded = c->b_is_folded ? 1 : 0;
    for (int k = 0; k < count; ++k) {
        if (ops[k].which_map != UVRT_MAP_SUM && ops[k].which_map != UVRT_MAP_MAX)
            return fail(UVRT_ERR_INVALID, "uvrt_replay_batch: which_map must be 0 or 1");
        p.ops[k].plane = c->b_phys[k];
        p.ops[k].duration = ops[k].duration;
        p.ops[k].shade = ops[k].shade;
        p.ops[k].which_map = ops[k].which_map;
        p.ops[k].photons_per_light = ops[k].photons_per_light;
        p.ops[k].scaled_power = ops[k].scaled_power;
        p.ops[k].min_value = ops[k].min_value;
        p.ops[k].threshold_view = ops[k].threshold_view;
    }
    launch_replay_batch(p, c->stream);
    HIP_TRY(hipGetLastError());
    if (tri_count < c->T) {     // a partial replay (calibration's 2-triangle scene never does this): clear the rest
        if (c->b_is_folded) HIP_TRY(hipMemsetAsync(c->bs[c->b_set].folded.p, 0, c->bs[c->b_set].folded.bytes, c->stream));
        else HIP_TRY(hipMemsetAsync(c->bs[c->b_set].planes.p, 0, c->bs[c->b_set].planes.bytes, c->stream));
    }
    HIP_TRY(hipEventRecord(c->bs[c->b_set].free_ev, c->stream));    // the set may be traced into again
    c->b_count = 0;
    c->b_is_folded = false;
    // later accumulate / Shade work waits for this replay; the next batch's generate / extend do not
    return mark_map_fence(c);
}

int uvrt_read_batch_counts(uvrt_ctx* c, int32_t launch, int32_t* out, int32_t first, int32_t count)
{
    if (!c || c->b_count <= 0 || launch < 0 || launch >= c->b_count)
        return fail(UVRT_ERR_INVALID, "uvrt_read_batch_counts: no such launch in the traced batch");
    if (int rc = uvrt_fold_batch(c)) return rc;
    if (!out || first < 0 || count < 0 || first + count > c->T) return fail(UVRT_ERR_INVALID, "uvrt_read_batch_counts: bad range");
    if (count == 0) return UVRT_OK;
    HIP_TRY(hipMemcpyAsync(out, c->bs[c->b_set].folded.as<int32_t>() + (size_t)c->b_phys[launch] * c->T + first, (size_t)count * 4,
                           hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UVRT_OK;
}


// ---------------------------------------------------------------- RCCL (one all-reduce per computation)
//
// librccl is opened lazily with dlopen: a process that never shards (the common case) does not load it,
// and one that already holds an RCCL (torch.distributed) gets that same library by its soname.
namespace {
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
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
    UVRT_SYM(AllReduce, ncclAllReduce);
    UVRT_SYM(GroupStart, ncclGroupStart);
    UVRT_SYM(GroupEnd, ncclGroupEnd);
    UVRT_SYM(GetErrorString, ncclGetErrorString);
#undef UVRT_SYM
    g_rccl.lib = h;
    return UVRT_OK;
}
#define RCCL_TRY(expr)                                                                         \
    do {                                                                                        \
        ncclResult_t r_ = (expr);                                                               \
        if (r_ != ncclSuccess)                                                                  \
            return fail(UVRT_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));        \
    } while (0)
}  // namespace

int uvrt_comm_unique_id(void* id128)
{
    if (!id128) return fail(UVRT_ERR_INVALID, "uvrt_comm_unique_id: null pointer");
    if (int rc = rccl_load()) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI hands the id over as 128 bytes");
    RCCL_TRY(g_rccl.GetUniqueId((ncclUniqueId*)id128));
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
    return UVRT_OK;
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
    return UVRT_OK;
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

int uvrt_advance_seed(uvrt_ctx* c, const float lp[3], float light_length)
{
    if (!c || !lp) return fail(UVRT_ERR_INVALID, "uvrt_advance_seed: null argument");
    c->seed = uvrt_seed_next_mode(lp, light_length, c->seed, c->seed_mode);
    return UVRT_OK;
}

int uvrt_set_wide_bvh(uvrt_ctx* c, int32_t on)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    c->wide = on != 0;
    return UVRT_OK;
}

int uvrt_set_hot_records(uvrt_ctx* c, int32_t mode)
{
    if (!c || (mode != 0 && mode != 1)) return fail(UVRT_ERR_INVALID, "uvrt_set_hot_records: mode must be 0 or 1");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    c->hot_mode = mode;
    c->recs_valid = false;
    for (int l = 0; l < uvrt_ctx::MAXL; ++l) { c->xrecs_valid[l] = false; c->lane_perm[l] = nullptr; }
    return UVRT_OK;
}

int uvrt_set_seed_mode(uvrt_ctx* c, int32_t mode)
{
    if (!c || (mode != 0 && mode != 1)) return fail(UVRT_ERR_INVALID, "uvrt_set_seed_mode: mode must be 0 or 1");
    c->seed_mode = mode;
    return UVRT_OK;
}

int uvrt_set_sort_bits(uvrt_ctx* c, int32_t bits)
{
    if (!c || bits < -1 || bits > 20) return fail(UVRT_ERR_INVALID, "uvrt_set_sort_bits: bits must be in [-1,20]");
    c->sort_bits = bits;
    return UVRT_OK;
}
int uvrt_set_record_hits(uvrt_ctx* c, int32_t on)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    c->record_hits = on != 0;
    return UVRT_OK;
}
int uvrt_set_flavour(uvrt_ctx* c, int32_t flavour)
{
    if (!c || (flavour != 0 && flavour != 1)) return fail(UVRT_ERR_INVALID, "uvrt_set_flavour: flavour must be 0 or 1");
    c->flavour = flavour;
    return UVRT_OK;
}
int uvrt_set_variant(uvrt_ctx* c, int32_t variant)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (!variant_ok(variant)) return fail(UVRT_ERR_INVALID, "uvrt_set_variant: %d is not a variant (0, 400-899)", variant);
    c->variant = variant;
    return UVRT_OK;
}
int uvrt_set_pipeline(uvrt_ctx* c, int32_t on)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    c->pipeline = on != 0;
    return UVRT_OK;
}
int uvrt_set_timing(uvrt_ctx* c, int32_t on)
{
    if (!c) return fail(UVRT_ERR_INVALID, "null context");
    c->timing = on != 0;
    return UVRT_OK;
}

int uvrt_read_rays(uvrt_ctx* c, void* rays32, int64_t first, int64_t count)
{
    if (!c || !rays32 || c->last_n < 0 || first < 0 || count < 0 || first + count > c->last_n)
        return fail(UVRT_ERR_INVALID, "uvrt_read_rays: range outside the last generate");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    if (count == 0) return UVRT_OK;
    if (int rc = c->export_buf.ensure((size_t)count * 32, false, c->stream)) return rc;
    const uint2* hits = (c->last_extended && c->hits.p) ? c->hits.as<uint2>() : nullptr;
    launch_export_rays(lane_rays(c).as<float4>(), hits, c->export_buf.p, c->ox, c->oz, first, count, c->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(rays32, c->export_buf.p, (size_t)count * 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return UVRT_OK;
}

int uvrt_write_rays(uvrt_ctx* c, const void* rays32, int64_t n)
{
    if (!c || !rays32 || n <= 0 || n > c->capacity)
        return fail(UVRT_ERR_INVALID, "uvrt_write_rays: n must be in (0, capacity]");
    if (int rc = set_device(c)) return rc;
    struct HostRay { float d[3], o[3], dist; uint32_t tri; };
    const HostRay* hr = (const HostRay*)rays32;
    std::vector<float> packed((size_t)n * 4);
    for (int64_t i = 0; i < n; ++i) {
        if (memcmp(&hr[i].o[0], &hr[0].o[0], 4) != 0 || memcmp(&hr[i].o[2], &hr[0].o[2], 4) != 0)
            return fail(UVRT_ERR_INVALID, "uvrt_write_rays: record %lld has a different orig.x/orig.z", (long long)i);
        packed[4 * i + 0] = hr[i].d[0]; packed[4 * i + 1] = hr[i].d[1];
        packed[4 * i + 2] = hr[i].d[2]; packed[4 * i + 3] = hr[i].o[1];
    }
    if (int rc = join_all(c)) return rc;
    c->lane = 0;
    c->cur_pipelined = false;
    HIP_TRY(hipMemcpyAsync(c->rays.p, packed.data(), (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->recs_valid = false;
    c->lane_perm[0] = nullptr;
    c->last_n = n;
    c->last_first = 0;
    c->last_sorted = false;
    c->last_extended = false;
    c->ox = hr[0].o[0];
    c->oz = hr[0].o[2];
    return UVRT_OK;
}

int uvrt_device_ptr(uvrt_ctx* c, int32_t which, void** ptr, int64_t* bytes)
{
    if (!c || !ptr || !bytes || !c->have_scene) return fail(UVRT_ERR_INVALID, "uvrt_device_ptr: bad argument");
    // the caller is about to touch the buffers on the context's stream: order it after the side lane,
    // and the side lane's next work after whatever the caller enqueues up to the next call
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    const DevBuf* b = nullptr;
    size_t elem = 0;
    switch (which) {
        case 0: b = &c->photon_map; elem = 8; break;
        case 1: b = &c->max_map; elem = 8; break;
        case 2:
            b = &lane_counts(c); elem = 4;
            launch_fold_counts(lane_counts(c).as<int32_t>(), c->replicas, c->T, c->T, c->stream);
            break;
        case 3: b = &c->dosage; elem = 4; break;
        case 4: b = &c->color; elem = 36; break;
        case 5:
            if (c->b_count <= 0) return fail(UVRT_ERR_INVALID, "uvrt_device_ptr: no traced batch");
            if (int rc = uvrt_fold_batch(c)) return rc;
            b = &c->bs[c->b_set].folded; elem = 4 * (size_t)c->b_count;
            break;
        default: return fail(UVRT_ERR_INVALID, "uvrt_device_ptr: which must be 0..5");
    }
    *ptr = b->p;
    *bytes = (int64_t)((size_t)c->T * elem);
    // see lane_stream(): the next side-lane work (on the maps: the next accumulate / Shade) waits for
    // what the caller enqueues on the main stream up to the next call
    // (the folded planes of a batch are only touched on the context's stream until their replay: no fence)
    if (which == 2) c->ext_touch = true; else if (which != 5) c->ext_touch_maps = true;
    return UVRT_OK;
}

int uvrt_copy_device(uvrt_ctx* c, int32_t which, void* ext, int32_t to_ctx)
{
    void* p = nullptr;
    int64_t bytes = 0;
    if (!ext) return fail(UVRT_ERR_INVALID, "uvrt_copy_device: null pointer");
    if (int rc = uvrt_device_ptr(c, which, &p, &bytes)) return rc;
    if (int rc = set_device(c)) return rc;
    if (to_ctx) HIP_TRY(hipMemcpyAsync(p, ext, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
    else HIP_TRY(hipMemcpyAsync(ext, p, (size_t)bytes, hipMemcpyDeviceToDevice, c->stream));
    return mark_fence(c);
}

int uvrt_extend_time_ms(uvrt_ctx* c, double* ms, int64_t* launches)
{
    if (!c || !ms || !launches) return fail(UVRT_ERR_INVALID, "null argument");
    if (int rc = set_device(c)) return rc;
    if (int rc = join_all(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    double total = 0;
    for (size_t i = 0; i < c->ev_used; ++i) {
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, c->ev_pool[i].first, c->ev_pool[i].second));
        total += t;
    }
    *ms = total;
    *launches = (int64_t)c->ev_used;
    c->ev_used = 0;
    return UVRT_OK;
}

}  // extern "C"
