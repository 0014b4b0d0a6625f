// uvrt_device.h -- device-side record layouts and kernel launch wrappers (gfx950 only).
//
// HBM layout of one context (see DESIGN.md "Data layout"):
//   pairs    : one 64-byte record per INNER node = the AABBs of its two children plus a 32-bit
//              reference to each child.  The reference's traversal always touches the two
//              children of a node together (extend.cl:56-59), so one aligned 64-byte record
//              (4 x dwordx4) replaces two 32-byte BVHNode gathers.
//   ltris    : one 48-byte record per triIdx slot (leaf order): v0, e1 = v1-v0, e2 = v2-v0 and
//              the original triangle id.  Replaces the triIdx[] -> Triangle[] double
//              indirection of extend.cl:50-53.
//   rays     : 16 bytes per photon {dir.xyz, orig.y}; orig.x / orig.z are the lamp's and are
//              launch-uniform (generate.cl:16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uvrt {

constexpr uint32_t REF_LEAF_BIT = 0x80000000u;   // bit 31: leaf reference
constexpr uint32_t REF_DONE = 0xFFFFFFFFu;       // traversal finished
constexpr uint32_t REF_FIRST_MASK = 0x07FFFFFFu; // leaf: first slot in ltris (27 bits)
constexpr int REF_COUNT_SHIFT = 27;              // leaf: 4-bit count code, 15 = see leaf_count[]
constexpr int MAX_TRIS = 1 << 27;

struct alignas(16) PairRec {   // 64 B
    float4 c0min_ref0;         // child0 min.xyz, bits of ref0
    float4 c0max_ref1;         // child0 max.xyz, bits of ref1
    float4 c1min;              // child1 min.xyz, 0
    float4 c1max;              // child1 max.xyz, 0
};

struct alignas(16) LeafTri {   // 48 B
    float4 v0_id;              // v0.xyz, bits of the original triangle id
    float4 e1;                 // v1 - v0 (one f32 subtraction per component, extend.cl:13)
    float4 e2;                 // v2 - v0
};

// 4-wide node of the opt-in collapsed tree (uvrt_extend4.hip): the boxes of up to four children -- the
// grandchildren of a node of the reference's BVH2 where its children are inner nodes -- and a reference to
// each.  128 bytes = two 64-byte record units; an empty slot holds a box no ray can hit and REF_DONE.
struct alignas(16) QuadRec {
    float4 xz[4];              // child k: min.x, max.x, min.z, max.z  (per launch: minus the lamp's x / z)
    float4 y01, y23;           // c0 min.y, c0 max.y, c1 min.y, c1 max.y ; likewise c2, c3
    uint32_t ref[4];           // inner: unit index (2 x node index); leaf: as in the BVH2 form, re-based per launch
    uint32_t pad[4];
};

struct SceneDev {
    const PairRec* pairs;
    const LeafTri* ltris;
    const uint32_t* leaf_count;  // per ltris slot: triangle count of the leaf starting there
    uint32_t root_ref;
    int32_t tri_count;
};

// One per-launch node-pair record of extend v6 (layout: uvrt_extend6.hip): the lamp's x / z
// subtracted from the x / z bounds (the single f32 subtraction of extend.cl:31,35), one axis of one
// child per register pair, leaf references re-based to record indices.
// `perm` (or nullptr = identity) renumbers the pair records: record i is written at perm[i] and the
// inner references are translated, so that the records the lamp's rays visit most form the prefix
// that the kernel serves from LDS.
__device__ __forceinline__ void prepare_record6(const PairRec* __restrict__ pairs, float4* __restrict__ recs,
                                                float ox, float oz, int32_t npairs, int src,
                                                const uint32_t* __restrict__ perm)
{
    const PairRec pr = pairs[src];
    uint32_t r0 = __float_as_uint(pr.c0min_ref0.w), r1 = __float_as_uint(pr.c0max_ref1.w);
    if (r0 >= REF_LEAF_BIT) r0 += (uint32_t)npairs; else if (perm) r0 = perm[r0];
    if (r1 >= REF_LEAF_BIT) r1 += (uint32_t)npairs; else if (perm) r1 = perm[r1];
    const int i = perm ? (int)perm[src] : src;
    recs[i * 4 + 0] = make_float4(pr.c0min_ref0.x - ox, pr.c0max_ref1.x - ox, pr.c0min_ref0.z - oz, pr.c0max_ref1.z - oz);
    recs[i * 4 + 1] = make_float4(pr.c1min.x - ox, pr.c1max.x - ox, pr.c1min.z - oz, pr.c1max.z - oz);
    recs[i * 4 + 2] = make_float4(pr.c0min_ref0.y, pr.c0max_ref1.y, pr.c1min.y, pr.c1max.y);
    recs[i * 4 + 3] = make_float4(__uint_as_float(r0), __uint_as_float(r1), 0.f, 0.f);
}

// ------------------------------------------------------------------ RNG, cl/tools.cl:2-4

__device__ __forceinline__ uint32_t wang_hash(uint32_t s)
{
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    return s;
}

__device__ __forceinline__ float random_float(uint32_t& s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return (float)s * 2.3283064365387e-10f;
}

// cl/generate.cl:11-37 for work-item `gid`: returns {dir.xyz, orig.y} (orig.x / orig.z are the lamp's).
// r0 = the first random float (position on the rod), (x, y) = the accepted disc sample: for the optional
// coherence key of k_generate.  seed_mode: include/uvrt.h uvrt_set_seed_mode.
__device__ __forceinline__ float4 generate_ray(float lx, float ly, float lz, float light_length, int64_t gid,
                                               uint32_t seed_prev, uint32_t seed_next, int32_t seed_mode,
                                               float& r0, double& x, double& y)
{
    const int threadID = (int)gid;                       // generate.cl:11 (int threadID)
    const uint32_t SEED = (gid == 0 || seed_mode == 1) ? seed_prev : seed_next;

    // generate.cl:13 -- f32 adds in source order, then float -> uint through int64
    float acc = (float)(threadID * 17 + 1);
    acc = acc + lx * 13.0f;
    acc = acc + ly * 7.0f;
    acc = acc + lz * 11.0f;
    acc = acc + (float)(SEED >> 15);
    uint32_t seed = wang_hash((seed_mode == 1 && acc < 0.0f) ? 0u : (uint32_t)(int64_t)acc);

    r0 = random_float(seed);
    const float origy = ly + r0 * light_length;          // :16
    const float diry = random_float(seed) * 2.0f - 1.0f; // :22
    const double dirxzlength = sqrt(1.0 - (double)diry * (double)diry);   // :23

    x = (double)(random_float(seed) * 2.0f - 1.0f);                       // :25
    y = (double)(random_float(seed) * 2.0f - 1.0f);
    while (x * x + y * y > 1.0) {                                         // :26-28
        x = (double)(random_float(seed) * 2.0f - 1.0f);
        y = (double)(random_float(seed) * 2.0f - 1.0f);
    }
    const double s = dirxzlength / sqrt(x * x + y * y);                   // :29
    return make_float4((float)(x * s), diry, (float)(y * s), origy);      // :31-37
}

// ---- the traversal kernel's LDS cache (uvrt_extend6.hip) ----
constexpr uint32_t TOP6_MAX = 175;        // records cached in LDS: 176 x 64 B + 9 stack rows = 20 KB per workgroup, eight per CU

// ---- batched tracing (include/uvrt.h uvrt_trace_batch): several launches' rays side by side ----
constexpr int MAX_BATCH = 64;          // launches per uvrt_trace_batch / uvrt_replay_batch call

struct GenBatchParams {
    float4* rays;                      // [count][n_pad]: physical plane p at rays + p * n_pad
    int64_t n_pad;                     // rays per plane in the buffer (n rounded up to 64)
    int64_t first_gid, n;              // global ids [first_gid, first_gid + n) of EVERY launch
    float light_length;
    int32_t seed_mode;
    int32_t count;
    float lx[MAX_BATCH], ly[MAX_BATCH], lz[MAX_BATCH];       // lamp of physical plane p
    uint32_t seed_prev[MAX_BATCH], seed_next[MAX_BATCH];     // its place in the SEED chain
};

// one entry of uvrt_replay_batch, in LOGICAL launch order
struct ReplayOp {
    int32_t plane;                     // physical plane of this launch
    float duration;                    // accumulate.cl timeStep
    int32_t shade;                     // run computeDosage + dosageToColor after this launch
    int32_t which_map;                 // 0 photonMap, 1 maxPhotonMap
    int32_t photons_per_light;
    float scaled_power, min_value;
    int32_t threshold_view;
};

struct ReplayParams {
    double* photon_map;
    double* max_map;
    int32_t* planes;                   // [plane][replicas][T] deposit replicas (folded == 0)
    int32_t* folded;                   // [plane][T] sums over the replicas (folded == 1)
    float* dosage;
    float* color;
    const float* area;
    int64_t plane_stride;              // ints between planes of `planes` (= replicas * T)
    int32_t replicas, T, count, is_folded;
    ReplayOp ops[MAX_BATCH];
};

struct GenParams {
    float4* rays;          // [n] gid order: dir.xyz, orig.y
    uint2* keyrank;        // [n] (key, rank within key) or nullptr when not sorting
    uint32_t* hist;        // [1 << sort_bits]
    float lx, ly, lz;      // lamp position (generate.cl arg 1)
    float light_length;    // generate.cl arg 2
    int64_t first_gid;
    int64_t n;
    uint32_t seed_prev;    // SEED_{k-1}: read by work-item 0
    uint32_t seed_next;    // SEED_k: read by everybody else
    int32_t seed_mode;     // 0 canonical (above); 1 "gfx950-ocl": every work-item reads SEED_{k-1} and a
                           // negative seed sum converts to 0 (include/uvrt.h uvrt_set_seed_mode)
    int32_t bits_phi, bits_y, bits_o;
    // extend v6's per-launch records, written by extra workgroups of the same launch (or nullptr)
    const PairRec* prep_pairs;
    float4* prep_recs;
    const uint32_t* prep_perm;
    int32_t prep_npairs;
    uint32_t ray_blocks;   // workgroups [0, ray_blocks) generate rays, the rest prepare records
};

struct ExtendParams {
    SceneDev scene;
    const float4* rays;      // [n] in trace order
    uint32_t chunk;          // persistent kernel: trace slots owned by each wavefront
    uint32_t* ovf_stack;     // persistent kernels: [grid threads][MAX_STACK - LDS entries] stack overflow
    uint64_t ovf_capacity;   // entries (uint32) available in ovf_stack; launches that need more are refused
    int32_t num_cus;         // compute units of the device (persistent grids are sized from it)
    uint32_t top_pairs;      // pair records [0, top_pairs) = the tree levels cached in LDS (<= TOP6_MAX)
    int32_t force_exact;     // scene or lamp position outside the fast path's proof conditions
    int32_t drain_merge;     // k_extend6: the four waves of a workgroup pool the last rays of their drains in one wave (merge6)
    int32_t flavour;         // 0 strict (canonical), 1 "ocl-amd" fused cross/dot in the triangle test
    const uint32_t* order;   // [n] trace slot -> local ray index, or nullptr (identity)
    uint2* hits;             // [n] by local ray index: (dist bits, triID), or nullptr
    int32_t* counts;         // tempPhotonMap, count_replicas copies count_stride ints apart: a
                             // workgroup deposits into copy (blockIdx % count_replicas), so hits on
                             // a hot triangle do not serialise on one address; accumulate folds them
    int32_t count_replicas;
    int64_t count_stride;
    uint32_t* error_flag;    // set to 1 on traversal stack overflow
    float ox, oz;            // launch-uniform origin components
    int64_t n;
    int32_t npairs;
    void* recs;              // extend v6: [npairs] per-launch pair records + [T] leaf records, 64 B each
    int32_t refill_min;      // extend v6: idle lanes that trigger a refill (16)
    const uint32_t* perm;    // extend v6: record renumbering (nullptr = identity), see prepare_record6
    int32_t recs_prepared;   // extend v6: recs[0, npairs) already hold this launch's records (k_generate)
    uint32_t root_ref6;      // root reference in v6's record numbering (set by launch_extend6)
    // batched tracing: `rays` holds nplanes planes of plane_batches * 64 slots each, of which the first
    // plane_n are rays; the deposits of plane k go to counts + k * plane_stride (+ replica * count_stride).
    // plane_batches == 0: one launch, n rays, no planes.
    uint32_t plane_batches;
    uint32_t plane_n;
    uint32_t plane_stride;
    float plane_inv;         // 1 / plane_batches (set by the launch wrapper): first guess of a batch's plane
    // the 4-wide form (uvrt_extend4.hip): per-launch records [2 * nquads units of 64 B] + leaf records
    const void* recs4;
    int32_t nquads;
    uint32_t top_quads;      // nodes [0, top_quads) are served from LDS
    uint32_t root_ref4;      // REF_DONE / a leaf reference (re-based) / 0
};

// launch wrappers (uvrt_kernels.hip)
void launch_generate(const GenParams& p, hipStream_t s);
void launch_generate_batch(const GenBatchParams& p, hipStream_t s);
// folded[p][i] = sum over the replicas of plane p, replicas zeroed (the all-reduce payload)
void launch_fold_planes(int32_t* planes, int32_t* folded, int32_t nplanes, int32_t replicas, int32_t T, hipStream_t s);
void launch_replay_batch(const ReplayParams& p, hipStream_t s);
void launch_add_counts(int32_t* dst, const int32_t* src, int64_t n, hipStream_t s);     // dst[i] += src[i]
void launch_prepare_launch6(const PairRec* pairs, void* recs, float ox, float oz, int32_t npairs, const uint32_t* perm,
                            hipStream_t s);
void launch_scan_bins(uint32_t* hist, uint32_t* bin_start, int32_t nbins, hipStream_t s);
void launch_scatter(const float4* rays, const uint2* keyrank, const uint32_t* bin_start,
                    float4* sorted, uint32_t* order, int64_t n, hipStream_t s);
// extend (uvrt_extend6.hip): code bits 0-1 = leaf period - 1, bit 2 = no LDS top cache; returns false
// (nothing launched) when the grid would not fit the overflow-stack buffer
bool launch_extend6(const ExtendParams& p, int code, int grid_per_cu, hipStream_t s);
// the opt-in 4-wide traversal (uvrt_extend4.hip)
bool launch_extend4(const ExtendParams& p, int grid_per_cu, hipStream_t s);
void launch_prepare_launch4(const QuadRec* quads, void* recs4, float ox, float oz, int32_t nquads, hipStream_t s);
// hot-record set-up (uvrt_hotset.hip): visit statistics -> selection -> renumbering for up to HS_GROUPS lamps at once
constexpr int HS_GROUPS = 16;
struct HotSetupParams {
    const PairRec* pairs;
    const LeafTri* ltris;
    const uint32_t* leaf_count;
    uint32_t root_ref;
    float light_length;
    int32_t seed_mode;
    int32_t n;                         // photons of the launch whose visits are counted (global ids [0, n))
    int32_t npairs, keep, count;       // records, hot records wanted, lamps
    int32_t tail_lanes;                // a wave of k_visit_stats stops once at most this many of its rays are under way
    int32_t direct_bins;               // k_select_hot looks for the hot records among the first direct_bins records before it walks the tree
    uint32_t* hist;                    // [count][npairs] visit counters: zero before, zero again after
    uint32_t* hot_list;                // [count][TOP6_MAX + 1] scratch: number of hot records + their indices, ascending
    uint32_t* perm[HS_GROUPS];         // out: the renumbering of lamp k, [npairs]
    float lx[HS_GROUPS], ly[HS_GROUPS], lz[HS_GROUPS];
    uint32_t seed_prev[HS_GROUPS], seed_next[HS_GROUPS];
};
void launch_hot_setup(const HotSetupParams& p, hipStream_t s);
void launch_prepare_leaves6(const LeafTri* ltris, void* recs, int32_t npairs, int32_t T, hipStream_t s);
constexpr uint64_t OVF_MAX_ENTRIES = (uint64_t)256 * 16 * 256 * 24;   // largest grid x deepest overflow
void launch_accumulate(double* photon_map, double* max_map, int32_t* counts, int32_t replicas,
                       int64_t stride, float time_step, int32_t T, hipStream_t s);
void launch_reset(double* photon_map, double* max_map, int32_t* counts, int32_t replicas,
                  int64_t stride, float* color, int32_t reset_color, int32_t T, hipStream_t s);
void launch_fold_counts(int32_t* counts, int32_t replicas, int64_t stride, int32_t T, hipStream_t s);
void launch_compute_dosage(const double* map, float* dosage, const float* area,
                           int32_t photons_per_light, float scaled_power, int32_t T,
                           hipStream_t s);
void launch_shade(const double* map, float* dosage, const float* area, float* color, int32_t photons_per_light,
                  float scaled_power, float min_value, int32_t threshold_view, int32_t T, hipStream_t s);
void launch_accumulate_shade(double* photon_map, double* max_map, int32_t* counts, int32_t replicas, int64_t stride,
                             float time_step, float* dosage, const float* area, float* color, int32_t which_map,
                             int32_t photons_per_light, float scaled_power, float min_value, int32_t threshold_view,
                             int32_t T, hipStream_t s);
void launch_dosage_to_color(const float* dosage, float* color, float min_value,
                            int32_t threshold_view, int32_t T, hipStream_t s);
void launch_prepare_scene(const float4* tris64, const uint32_t* tri_idx, LeafTri* ltris,
                          float* area, int32_t T, hipStream_t s);
void launch_clock_probe(unsigned long long* out2, unsigned long long ticks_100mhz, hipStream_t s);
void launch_export_rays(const float4* rays, const uint2* hits, void* out32, float ox, float oz,
                        int64_t first, int64_t count, hipStream_t s);

}  // namespace uvrt
