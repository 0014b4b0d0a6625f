/*
 * uvrt_oracle.h -- CPU restatement of the reference UV-dose hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (the package directory, include/, the
 * C-ABI library) may include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout).  Arithmetic is the "strict" flavour SURVEY.md section 8c declares
 * canonical: IEEE-754 binary32/binary64, source order, no FMA contraction, OpenCL
 * min(x,y) = y<x?y:x, max(x,y) = x<y?y:x, dot = a.x*b.x + a.y*b.y (+ a.z*b.z),
 * length = sqrtf(x*x + y*y + z*z).
 *
 * Pinning: the reference has no tests.  The pins are the probe outputs
 * of the reference's own kernels recorded in SURVEY.md section 8c (tests/golden/survey_8c.json):
 * triangle count, floor height, BVH census, hit counts, dose[0..7], dose sums, SEED chain.
 */
#ifndef UVRT_ORACLE_H
#define UVRT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cl/tools.cl:8-14 -- 32 bytes */
typedef struct {
    float dirx, diry, dirz;
    float origx, origy, origz;
    float dist;
    uint32_t triID;
} orc_ray;

/* cl/tools.cl:31-37 == mesh.h:6-13 -- 64 bytes */
typedef struct {
    float v0x, v0y, v0z, dummy1;
    float v1x, v1y, v1z, dummy2;
    float v2x, v2y, v2z, dummy3;
    float cx, cy, cz, dummy4;
} orc_tri;

/* cl/tools.cl:39-45 == bvh.h:11-21 -- 32 bytes */
typedef struct {
    float minx, miny, minz;
    int32_t leftFirst;
    float maxx, maxy, maxz;
    int32_t triCount;
} orc_node;

/* cl/tools.cl:16-21 -- 36 bytes */
typedef struct {
    float v0x, v0y, v0z, v1x, v1y, v1z, v2x, v2y, v2z;
} orc_tricolor;

/* per-launch traversal census, feeds the algorithmic-bytes formula of SURVEY.md section 8d */
typedef struct {
    uint64_t rays;
    uint64_t node_visits;   /* iterations of the traversal loop that looked at a node */
    uint64_t aabb_tests;    /* calls of IntersectAABB */
    uint64_t tri_tests;     /* calls of IntersectTri */
    uint64_t hits;          /* rays with dist != 1e30f */
    uint32_t max_stack;     /* deepest stackPtr seen */
    uint32_t* visit_hist;   /* optional: visits per node index (analysis helper), NULL otherwise */
} orc_stats;

/* cl/tools.cl:2-4 */
uint32_t orc_wang_hash(uint32_t s);
uint32_t orc_random_int(uint32_t* s);
float    orc_random_float(uint32_t* s);

/* cl/generate.cl:13 -- the seed expression, strict f32 source order, float->uint via int64 */
uint32_t orc_seed_of(int32_t tid, const float lightPos[3], uint32_t SEED);

/* cl/generate.cl:8-40 for ONE work-item; returns the work-item's final RNG state */
uint32_t orc_generate_one(orc_ray* out, int32_t tid, const float lightPos[3], float lightLength,
                          uint32_t SEED);

/* cl/generate.cl:8-40 for gids [first, first+n) under the pinned SEED semantics of SURVEY.md
 * section 8c: work-item 0 reads *SEED (= SEED_{k-1}) and stores SEED_k; every other work-item
 * reads SEED_k.  rays[i] receives gid first+i.  *SEED is advanced to SEED_k (also when the
 * range does not contain gid 0: SEED_k is a pure function of lightPos and SEED_{k-1}). */
void orc_generate(orc_ray* rays, int64_t first, int64_t n, const float lightPos[3],
                  float lightLength, uint32_t* SEED);

/* Test helper: every work-item of [first, first+n) reads the SAME SEED (one outcome of the
 * reference's SEED race on a GPU, generate.cl:6,13,39); saturate != 0: a negative seed sum becomes 0
 * (v_cvt_u32_f32) instead of taking the int64 route.  Returns gid 0's final RNG state (or 0). */
uint32_t orc_generate_fixed_seed(orc_ray* rays, int64_t first, int64_t n, const float lightPos[3],
                                 float lightLength, uint32_t SEED, int saturate);

/* Arithmetic flavour of IntersectTri's cross()/dot() (extend.cl:14-24):
 *   0 (default, canonical -- SURVEY.md 8c): unfused, dot = x*x + y*y + z*z, cross = a*b - c*d
 *   1 "ocl-amd": what the reference's extend.cl becomes when built with ROCm's OpenCL device
 *     library for gfx950 (oracle/_ref): cross(a,b).x = fma(a.y, b.z, -(a.z*b.y)) (and cyclic),
 *     dot(a,b) = fma(a.z, b.z, fma(a.y, b.y, a.x*b.x)); read off the disassembly of
 *     oracle/_ref/ref_extend.co.  Used to compare bit for bit with those kernels on the GPU.
 *   2 "shipped flags": what the reference's extend.cl becomes with the reference's OWN build flags
 *     (-cl-fast-relaxed-math -cl-mad-enable -cl-single-precision-constant, template/template.cpp:1192) on gfx950,
 *     read off the disassembly of oracle/_ref/ref_extend_fast.co: IntersectAABB's t = (b - o) * v_rcp_f32(d) with the
 *     hardware's min / max, IntersectTri as flavour 1 with f = v_rcp_f32(a).  v_rcp_f32 is a hardware approximation:
 *     the oracle evaluates it through a table read from the GPU (orc_set_rcp_table, rcp_model.h), so this flavour
 *     exists only where a gfx950 is at hand; without a table orc_extend must not be called in flavour 2. */
void orc_set_flavour(int flavour);
/* 2^23 entries: bits of v_rcp_f32(1.m) per significand m (oracle/rcp_probe.hip:refgpu_rcp_table); the pointer is kept */
void orc_set_rcp_table(const uint32_t* table23);
int orc_have_rcp_table(void);
float orc_rcp_model(float x);

/* analysis helper: node visits per ray (rays untouched) */
void orc_extend_steps(const orc_tri* tris, const orc_ray* rays, int64_t n, const orc_node* nodes,
                      const uint32_t* triIdx, uint16_t* steps);

void orc_extend_visit_hist(const orc_tri* tris, const orc_ray* rays, int64_t n, const orc_node* nodes,
                           const uint32_t* triIdx, uint32_t* hist);

/* cl/extend.cl:85-99 over n rays (OpenMP over rays; counts are order independent).
 * stats may be NULL. nthreads <= 0 -> OpenMP default. */
void orc_extend(int32_t* tempPhotonMap, const orc_tri* tris, orc_ray* rays, int64_t n,
                const orc_node* nodes, const uint32_t* triIdx, orc_stats* stats, int nthreads);

/* cl/accumulate.cl:4-14 */
void orc_accumulate(double* photonMap, double* maxPhotonMap, int32_t* tempPhotonMap,
                    float timeStep, int32_t T);
/* cl/reset.cl:4-26 */
void orc_reset(double* photonMap, double* maxPhotonMap, int32_t* tempPhotonMap,
               orc_tricolor* colorMap, int32_t resetColor, int32_t T);
/* cl/shade.cl:23-41 */
void orc_compute_dosage(const double* photonMap, float* dosageMap, const orc_tri* tris,
                        int32_t photonsPerLight, float scaledPower, int32_t T);
/* cl/shade.cl:43-71 (+ :4-21) */
void orc_dosage_to_color(const float* dosageMap, orc_tricolor* colorMap, float minValue,
                         int32_t thresholdView, int32_t T);

/* mesh.cpp:100-136; yvals = the y coordinate of every (duplicated) vertex, 3 per triangle */
float orc_floor_height(const float* yvals, int32_t count);

/* bvh.cpp:5-220 (SSE path semantics, serial; numbering is thread-count independent because
 * sub-tree node ranges are pre-reserved, bvh.cpp:33-42).  Writes tris[i].c{x,y,z}
 * (bvh.cpp:23).  nodes must hold nodes_cap >= 2*T+64 entries (the reference's 2T pool
 * overflows); returns the true extent (highest written index + 1), or -1. */
int32_t orc_bvh_build(orc_tri* tris, int32_t T, orc_node* nodes, int32_t nodes_cap,
                      uint32_t* triIdx);

#ifdef __cplusplus
}
#endif
#endif
