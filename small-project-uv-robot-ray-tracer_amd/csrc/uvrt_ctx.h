// uvrt_ctx.h -- the context behind the C ABI of include/uvrt.h, shared by the translation units that implement it:
//   uvrt_capi.hip         context, scene, buffers, knobs, read-backs and test hooks
//   uvrt_capi_launch.hip  the per-launch entry points (generate, extend, accumulate, shade) and the launch lanes
//   uvrt_capi_batch.hip   batched tracing (uvrt_trace_batch / fold / replay)
//   uvrt_capi_comm.hip    the one collective of a sharded computation (RCCL, opened at run time)
//
// One context = one HIP device + one in-order stream + all device buffers of a RayTracer
// (raytracer.h:50-53).  There is no CPU fallback: every entry point either runs on the GPU or
// returns an error.
#pragma once
#include "../../include/uvrt.h"
#include "uvrt_device.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace uvrt_impl {

extern thread_local std::string g_err;      // uvrt_last_error()
int fail(int code, const char* fmt, ...);   // sets g_err, returns code

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return ::uvrt_impl::fail(UVRT_ERR_HIP, "%s failed: %s (%s:%d)", #expr,      \
                                     hipGetErrorString(e_), __FILE__, __LINE__);        \
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

}  // namespace uvrt_impl

using uvrt_impl::DevBuf;

struct uvrt_ctx {
    int device = 0;
    int num_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;

    // scene
    int32_t T = 0;
    DevBuf pairs, recs, perm, ltris, leaf_count, area;
    bool have_perm = false;      // the caller's own record renumbering (uvrt_set_record_perm)
    uint64_t perm_clock = 0;     // stamps every renumbering written into `perm` or a hot entry
    uint64_t perm_gen = 0;       // stamp of what `perm` holds
    int32_t npairs = 0;
    uint32_t root_ref = uvrt::REF_DONE;
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
    bool drain_merge = true;                   // k_extend6's workgroups pool the last rays of their waves (ExtendParams::drain_merge; developer knob UVRT_DRAIN_MERGE=0)
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
                                      // per CU) win only for long launch sequences (profiles/r01/r01_v6_experiments.txt)
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
    // The renumberings live in slabs of HOT_SLAB entries (the first one allocated with the scene, so that a new lamp
    // costs no allocation); the visit counters and the hot list are scratch of the three set-up kernels, one set per
    // launch lane (the kernels of one lamp run back to back on one stream and leave the counters zeroed).
    static constexpr int HOT_SLAB = 16, HOT_MAX = 64;
    struct HotEntry { float lamp[3]; uint32_t* perm; uint64_t stamp; hipEvent_t ready; uint64_t gen; };
    std::vector<HotEntry> hot;
    std::vector<DevBuf> hot_slabs;        // [ceil(entries / HOT_SLAB)]: HOT_SLAB x npairs uint32 each
    DevBuf hot_hist[MAXL], hot_list[MAXL];
    int32_t hot_sample = 32768;           // photons of the launch whose visits are counted (developer knob UVRT_HOT_SAMPLE)
    int32_t hot_direct = 8192;            // records k_select_hot takes as candidates without a tree walk (developer knob UVRT_HOT_DIRECT; tests force the walk with it)
    int32_t hot_tail = 16;                // straggler rule of k_visit_stats (developer knob UVRT_HOT_TAIL, uvrt_hotset.hip)
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
    int batch_lanes = 2;                  // side lanes the chunks of a batch alternate over (developer knob UVRT_BATCH_LANES: 1..3)
    int32_t b_repl = 16;                  // deposit replicas per plane of the traced batch
    std::vector<DevBuf> b_recs;           // [group]
    // `gen` tells two renumberings apart that live at ONE address: the caller's buffer after another
    // uvrt_set_record_perm, a hot entry recycled for another lamp (perm_clock stamps every (re)written renumbering)
    struct RecsKey { float ox = 0, oz = 0; const uint32_t* perm = nullptr; uint64_t gen = 0; bool valid = false; };
    std::vector<RecsKey> b_recs_key;      // what b_recs[g] holds
    int32_t b_count = 0;                  // launches of the batch that has not been replayed (0: none)
    int64_t b_n = 0, b_npad = 0;
    int32_t b_phys[uvrt::MAX_BATCH] = {};       // logical launch -> physical plane (launches are grouped by lamp)
    bool b_is_folded = false;             // b_folded holds the batch (fold / all-reduce done), the replicas are zero
    void* comm = nullptr;                 // ncclComm_t of a ray-range-sharded job (uvrt_comm_init_rank)
    // CUs the launch lanes leave to the context's stream while a communicator is set (uvrt_capi_comm.hip
    // reserve_cus_for_comm): the lanes' streams carry a CU mask, the persistent grids are sized for the rest
    int comm_reserve_knob = 8;            // developer knob UVRT_COMM_RESERVE_CUS (a multiple of 8, 0 = none)
    int lanes_masked_cus = 0;             // CUs masked out of the side lanes' streams right now
    int comm_rank = 0, comm_world = 1;

    // generate.cl:6 program-scope SEED
    uint32_t seed = 0;
    int32_t seed_mode = 0;   // uvrt_set_seed_mode

    // knobs
    int32_t sort_bits = 0;   // ray ordering off by default: extend is VALU-bound
    bool record_hits = false;
    int32_t variant = 0;
    int32_t flavour = 0;
#ifdef UVRT_DEV_VARIANTS
    // timing-only probe of the developer build (UVRT_PROBE_SKIP_GENERATE=n): after n uvrt_trace_batch calls the generate
    // launches are skipped -- the buffers still hold the same rays when every computation is the same (bench.py), so
    // the dose stays right and the step shows what k_generate_batch costs the pipeline (an upper bound for any scheme
    // that makes the rays elsewhere)
    int probe_skip_generate = 0;
    int probe_batches = 0;
#endif
    size_t batch_chunk_bytes = (size_t)96 << 20;   // rays per fused launch of a batch (developer knob UVRT_BATCH_CHUNK_MB,
                                                   // read once in uvrt_create): a chunk's rays stay in the Infinity Cache

    // A deferred accumulate (uvrt_accumulate stores it, its ordering already enqueued on the lane's stream): the Shade that
    // the host loop runs right after a launch (myapp.cpp:159-160) takes it along in ONE kernel (k_accumulate_shade); every
    // other entry point first launches it as the plain k_accumulate (set_device -> flush_pending).  Same arithmetic, same
    // order of operations on the maps: the launch count of an iteration drops from 4 to 3.
    struct PendingAcc { bool valid = false; int lane = 0; float time_step = 0; } pend;
    // traversal error flag + extend timing
    uint32_t* host_flag = nullptr;             // pinned, device-visible host word the kernels raise on a stack overflow: uvrt_sync
    uint32_t* host_flag_dev = nullptr;         // reads it after the stream sync, no copy (its device-side address)
    DevBuf error_flag;                         // (developer build with trip statistics: the flag and the counters behind it)
    hipStream_t probe_stream = nullptr;        // uvrt_clock_probe_start's own stream (created on first use)
    DevBuf probe_out;                          // {shader ticks, 100 MHz ticks}
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
};

namespace uvrt_impl {
using namespace uvrt;

// ---- launch lanes ----
inline hipStream_t stream_of(uvrt_ctx* c, int l) { return l == 0 ? c->stream : c->side[l]; }

inline int set_device_only(uvrt_ctx* c)
{
    HIP_TRY(hipSetDevice(c->device));
    return UVRT_OK;
}
// the deferred accumulate as a launch of its own, on the stream of the lane it belongs to (ordering enqueued by uvrt_accumulate)
inline int flush_pending(uvrt_ctx* c)
{
    if (!c->pend.valid) return UVRT_OK;
    c->pend.valid = false;
    const int l = c->pend.lane;
    launch_accumulate(c->photon_map.as<double>(), c->max_map.as<double>(), (l ? c->xcounts[l] : c->counts).as<int32_t>(),
                      c->replicas, c->T, c->pend.time_step, c->T, stream_of(c, l));
    HIP_TRY(hipGetLastError());
    return UVRT_OK;
}
// every entry point that touches the device starts here (uvrt_shade, which may take a deferred accumulate along, uses
// set_device_only)
inline int set_device(uvrt_ctx* c)
{
    HIP_TRY(hipSetDevice(c->device));
    return flush_pending(c);
}
// the main stream becomes ordered after everything the side streams hold
inline int join_all(uvrt_ctx* c)
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
inline int mark_fence(uvrt_ctx* c)
{
    HIP_TRY(hipEventRecord(c->ev_fence, c->stream));
    ++c->fence_seq;
    return UVRT_OK;
}
// an operation on the per-triangle maps (reset, an external reduction) has been enqueued on the main
// stream: later accumulate / Shade work on side streams waits for it -- generate and extend do not,
// so the first launches of the next computation overlap the drain of the previous one
inline int mark_map_fence(uvrt_ctx* c)
{
    HIP_TRY(hipEventRecord(c->ev_mapfence, c->stream));
    ++c->mapfence_seq;
    return UVRT_OK;
}
// stream of the current lane; a side stream first catches up with the last context-wide operation
// and, for work on the maps (`maps`), with the last operation on them
inline int lane_stream(uvrt_ctx* c, hipStream_t* out, bool maps = false)
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
inline int order_after_previous(uvrt_ctx* c)
{
    const int l = c->lane, q = c->prev_lane;
    if (q == l) return UVRT_OK;
    if (l == 0) return join_all(c);
    HIP_TRY(hipEventRecord(c->ev_tail[q], stream_of(c, q)));
    HIP_TRY(hipStreamWaitEvent(c->side[l], c->ev_tail[q], 0));
    c->side_used[l] = true;
    return UVRT_OK;
}
// compute units the persistent grid of a launch on the current lane is sized for
inline int lane_cus(const uvrt_ctx* c) { return c->lane == 0 ? c->num_cus : c->num_cus - c->lanes_masked_cus; }
inline DevBuf& lane_rays(uvrt_ctx* c) { return c->lane ? c->xrays[c->lane] : c->rays; }
inline DevBuf& lane_recs(uvrt_ctx* c) { return c->lane ? c->xrecs[c->lane] : c->recs; }
inline DevBuf& lane_counts(uvrt_ctx* c) { return c->lane ? c->xcounts[c->lane] : c->counts; }
inline DevBuf& lane_ovf(uvrt_ctx* c) { return c->lane ? c->xovf[c->lane] : c->ovf_stack; }

// work-item 0's RNG walk of cl/generate.cl:13-39 on the host (strict f32/f64, same order)
inline uint32_t host_wang_hash(uint32_t s)
{
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    return s;
}
inline float host_random_float(uint32_t& s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return (float)s * 2.3283064365387e-10f;
}

inline void split_bits(int bits, int& bphi, int& by, int& bo)
{
    bo = bits / 4;
    bphi = (bits - bo + 1) / 2;
    by = bits - bo - bphi;
}

// Kernel knobs (uvrt_set_variant).  0 (default) = the top-of-tree LDS cache, leaf visits every second
// trip, refill at 8 idle lanes, 8 workgroups per CU; 400-499 = code + 10 * grid code (uvrt_extend6.hip:
// code bits 0-1 leaf period - 1, bit 2 no LDS cache; grid code 0..4 = 8 / 4 / 6 / 2 / 16 workgroups per CU)
// with refill at 16 idle lanes; 500-599 = the same with IEEE divisions everywhere; 600-1299 = like 400-499 with the refill threshold 8 / 24 / 4 / 32 / 40 / 48 / 56 idle
// lanes (hundreds digit 6 ... 12).  (The v1-v5 kernels of round 1 are gone: see git history
// and DESIGN.md section 4 for what they measured.)
#ifdef UVRT_DEV_VARIANTS
inline bool variant_ok(int v) { return v == 0 || (v >= 400 && v < 1300); }
#else      // the product library holds the default kernel only: code 1 (leaf period 2, LDS cache) with any grid / refill knob
inline bool variant_ok(int v) { return v == 0 || (v >= 400 && v < 1300 && v % 10 == 1); }
#endif

inline int auto_sort_bits(int64_t n)
{
    // about one wave (64 rays) per key; no ordering for launches too small to matter
    if (n < 8192) return 0;
    int b = 0;
    while ((int64_t(64) << (b + 1)) <= n && b < 18) ++b;
    return b;
}


inline bool variant_is_knob(int v) { return v >= 400 && v < 1300; }
// idle lanes that trigger a refill.  Default 8; 24 for scenes of a million records and more, where a trip is a miss to the
// fabric whatever its lanes do and fewer, fuller refills are worth 2-3 % (profiles/r03/r03_soup_knobs.txt; the room is flat
// between 8 and 16 and loses at 24)
inline int variant_refill_min(int v, size_t records)
{
    if (!variant_is_knob(v)) return records >= ((size_t)1 << 20) ? 24 : 8;
    return v >= 1200 ? 56 : v >= 1100 ? 48 : v >= 1000 ? 40 : v >= 900 ? 32 : v >= 800 ? 4 : v >= 700 ? 24 : v >= 600 ? 8 : 16;
}
inline int variant_code6(int v) { return !variant_is_knob(v) ? 1 : v % 10; }   // default: LDS top cache, leaf period 2
inline int variant_per_cu(int v, int dflt)
{
    static const int per_cu[6] = {8, 4, 6, 2, 16, 7};
    const int gcode = (v / 10) % 10;
    return !variant_is_knob(v) ? dflt : per_cu[gcode < 6 ? gcode : 0];
}
// The record renumbering for a launch from `lamp` on launch lane `lane` (stream `s`): the caller's own
// (uvrt_set_record_perm), the automatic hot-record one (uvrt_hotset.hip: three small kernels enqueued on `s` the
// first time the lamp is seen), or none.  (uvrt_capi_launch.hip)
int launch_perm(uvrt_ctx* c, const float lamp[3], float light_length, uint32_t seed_prev, uint32_t seed_next,
                hipStream_t s, int lane, const uint32_t** out);
// the two halves of launch_perm: the cached renumbering of `lamp` (*out), or a reserved entry (*fresh, its `perm` pointer
// already final) whose set-up hot_build enqueues on `s` -- for several new lamps at once in ONE launch of each kernel
int hot_lookup(uvrt_ctx* c, const float lamp[3], hipStream_t s, const uint32_t** out, uvrt_ctx::HotEntry** fresh);
int hot_build(uvrt_ctx* c, uvrt_ctx::HotEntry* const* entries, const uint32_t* seed_prev, const uint32_t* seed_next, int count,
              float light_length, hipStream_t s, int lane);
// stamp of the renumbering at `perm` (0: none): what a key of prepared records remembers beside the address
inline uint64_t perm_generation(const uvrt_ctx* c, const uint32_t* perm)
{
    if (!perm) return 0;
    if (perm == c->perm.as<uint32_t>()) return c->perm_gen;
    for (const auto& h : c->hot) if (h.perm == perm) return h.gen;
    return 0;
}
// drops every cached renumbering (a new scene); with `slab`, the first slab of the new scene is allocated at once
int hot_reset(uvrt_ctx* c, bool slab);
// (re)creates the side lanes' streams with `reserve` CUs masked out, one per XCD and mask word of 8 (0: plain streams)
int set_lane_cu_mask(uvrt_ctx* c, int reserve);
// the scene part of an ExtendParams
inline void fill_scene(const uvrt_ctx* c, ExtendParams& p)
{
    p.scene.pairs = c->pairs.as<PairRec>();
    p.scene.ltris = c->ltris.as<LeafTri>();
    p.scene.leaf_count = c->leaf_count.as<uint32_t>();
    p.scene.root_ref = c->root_ref;
    p.scene.tri_count = c->T;
}

}  // namespace uvrt_impl
