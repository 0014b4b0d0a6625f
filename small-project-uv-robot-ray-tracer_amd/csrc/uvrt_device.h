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
    uint32_t top_pairs;      // pair records [0, top_pairs) = the tree levels cached in LDS (<= 127)
    int32_t force_exact;     // scene or lamp position outside the fast path's proof conditions
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
    uint32_t perm_root;      // perm[0]
    int32_t recs_prepared;   // extend v6: recs[0, npairs) already hold this launch's records (k_generate)
    uint32_t root_ref6;      // root reference in v6's record numbering (set by launch_extend6)
};

// launch wrappers (uvrt_kernels.hip)
void launch_generate(const GenParams& p, hipStream_t s);
void launch_scan_bins(uint32_t* hist, uint32_t* bin_start, int32_t nbins, hipStream_t s);
void launch_scatter(const float4* rays, const uint2* keyrank, const uint32_t* bin_start,
                    float4* sorted, uint32_t* order, int64_t n, hipStream_t s);
// extend (uvrt_extend6.hip): code bits 0-1 = leaf period - 1, bit 2 = no LDS top cache; returns false
// (nothing launched) when the grid would not fit the overflow-stack buffer
bool launch_extend6(const ExtendParams& p, int code, int grid_per_cu, hipStream_t s);
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
void launch_dosage_to_color(const float* dosage, float* color, float min_value,
                            int32_t threshold_view, int32_t T, hipStream_t s);
void launch_prepare_scene(const float4* tris64, const uint32_t* tri_idx, LeafTri* ltris,
                          float* area, int32_t T, hipStream_t s);
void launch_export_rays(const float4* rays, const uint2* hits, void* out32, float ox, float oz,
                        int64_t first, int64_t count, hipStream_t s);

}  // namespace uvrt
