/*
 * uvrt_oracle.c -- CPU restatement of the reference UV-dose hot path (see uvrt_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path and the timed CPU baseline.
 * Build with: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp (oracle/Makefile).
 * x86-64 SSE2 scalar arithmetic is IEEE binary32/binary64 with no excess precision, so each
 * C operator below is exactly one rounding, in the order the reference source writes it.
 */
#include "uvrt_oracle.h"
#include "rcp_model.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- RNG: cl/tools.cl:2-4 */

uint32_t orc_wang_hash(uint32_t s)
{
    s = (s ^ 61u) ^ (s >> 16);
    s *= 9u;
    s = s ^ (s >> 4);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15);
    return s;
}

uint32_t orc_random_int(uint32_t* s)
{
    *s ^= *s << 13;
    *s ^= *s >> 17;
    *s ^= *s << 5;
    return *s;
}

float orc_random_float(uint32_t* s)
{
    /* uint -> float (round to nearest even), then one f32 multiply by 2^-32 */
    return (float)orc_random_int(s) * 2.3283064365387e-10f;
}

/* ------------------------------------------------------------ generate: cl/generate.cl */

/* cl/generate.cl:13.  C usual-arithmetic conversions give
 *   ((((float)(tid*17+1) + lp.x*13.0f) + lp.y*7.0f) + lp.z*11.0f) + (float)(SEED>>15)
 * followed by an implicit float->uint conversion, taken through int64 (SURVEY.md 8a/8c:
 * what the x86-64 build of the reference kernel does; negative sums wrap modulo 2^32). */
uint32_t orc_seed_of(int32_t tid, const float lp[3], uint32_t SEED)
{
    float acc = (float)(tid * 17 + 1);
    acc = acc + lp[0] * 13.0f;
    acc = acc + lp[1] * 7.0f;
    acc = acc + lp[2] * 11.0f;
    acc = acc + (float)(SEED >> 15);
    return (uint32_t)(int64_t)acc;
}

static uint32_t generate_from_hash_input(orc_ray* out, uint32_t hash_input, const float lp[3],
                                         float lightLength)
{
    uint32_t seed = orc_wang_hash(hash_input);                           /* :13 */

    out->origx = lp[0];                                                  /* :16-19 */
    out->origy = lp[1] + orc_random_float(&seed) * lightLength;
    out->origz = lp[2];

    float diry = orc_random_float(&seed) * 2.0f - 1.0f;                  /* :22 */
    double dirxzlength = sqrt(1.0 - (double)diry * (double)diry);        /* :23 */

    /* :25 -- vector literal elements are evaluated left to right */
    double x = (double)(orc_random_float(&seed) * 2.0f - 1.0f);
    double y = (double)(orc_random_float(&seed) * 2.0f - 1.0f);
    while (x * x + y * y > 1.0) {                                        /* :26-28 */
        x = (double)(orc_random_float(&seed) * 2.0f - 1.0f);
        y = (double)(orc_random_float(&seed) * 2.0f - 1.0f);
    }
    double s = dirxzlength / sqrt(x * x + y * y);                        /* :29 */
    x = x * s;
    y = y * s;

    out->dirx = (float)x;                                                /* :31-35 */
    out->diry = diry;
    out->dirz = (float)y;
    out->dist = 1e30f;
    out->triID = 0;
    return seed;                                                         /* :39 (tid 0) */
}

uint32_t orc_generate_one(orc_ray* out, int32_t tid, const float lp[3], float lightLength,
                          uint32_t SEED)
{
    return generate_from_hash_input(out, orc_seed_of(tid, lp, SEED), lp, lightLength);
}

/* Test helper for pinning generate.cl against the reference's own kernel running on a GPU, where
 * SEED is racy (generate.cl:6,13,39): EVERY work-item of [first, first+n) reads the same SEED.
 * saturate != 0 converts a negative seed sum to 0 instead of going through int64: float -> uint of
 * a negative value is undefined in OpenCL C, and v_cvt_u32_f32 (what the reference's kernel becomes
 * on gfx950) saturates.  Returns the final RNG state of gid 0 (what it would store in SEED), or 0
 * when gid 0 is not in the range. */
uint32_t orc_generate_fixed_seed(orc_ray* rays, int64_t first, int64_t n, const float lp[3],
                                 float lightLength, uint32_t SEED, int saturate)
{
    uint32_t seed0 = 0;
    for (int64_t i = 0; i < n; i++) {
        const int32_t tid = (int32_t)(first + i);
        uint32_t h = orc_seed_of(tid, lp, SEED);
        if (saturate) {
            float acc = (float)(tid * 17 + 1);
            acc = acc + lp[0] * 13.0f;
            acc = acc + lp[1] * 7.0f;
            acc = acc + lp[2] * 11.0f;
            acc = acc + (float)(SEED >> 15);
            if (acc < 0.0f) h = 0u;
        }
        const uint32_t fin = generate_from_hash_input(&rays[i], h, lp, lightLength);
        if (first + i == 0) seed0 = fin;
    }
    return seed0;
}

void orc_generate(orc_ray* rays, int64_t first, int64_t n, const float lp[3],
                  float lightLength, uint32_t* SEED)
{
    /* SEED_k = final RNG state of work-item 0, which itself read SEED_{k-1} */
    orc_ray scratch;
    const uint32_t seed_prev = *SEED;
    const uint32_t seed_next = orc_generate_one(&scratch, 0, lp, lightLength, seed_prev);
    for (int64_t i = 0; i < n; i++) {
        int64_t gid = first + i;
        orc_generate_one(&rays[i], (int32_t)gid, lp, lightLength,
                         gid == 0 ? seed_prev : seed_next);
    }
    *SEED = seed_next;
}

/* ---------------------------------------------------------------- extend: cl/extend.cl */

static inline float cl_minf(float x, float y) { return y < x ? y : x; }
static inline float cl_maxf(float x, float y) { return x < y ? y : x; }

static int g_flavour = 0;
void orc_set_flavour(int flavour) { g_flavour = flavour; }

/* flavour 2 ("shipped flags", uvrt_oracle.h): v_rcp_f32 through the table measured on the GPU (rcp_model.h) */
static const uint32_t* g_rcp_table = NULL;
void orc_set_rcp_table(const uint32_t* table23) { g_rcp_table = table23; }
int orc_have_rcp_table(void) { return g_rcp_table != NULL; }
static inline float hw_rcp(float x)
{
    union { float f; uint32_t u; } a, r;
    a.f = x;
    r.u = orc_rcp_model_bits(a.u, g_rcp_table);
    return r.f;
}
float orc_rcp_model(float x) { return g_rcp_table ? hw_rcp(x) : NAN; }
/* v_min_f32 / v_max_f32 in IEEE mode: a NaN operand yields the other one; -0 < +0 */
static inline float hw_minf(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? a : b;
    return a < b ? a : b;
}
static inline float hw_maxf(float a, float b)
{
    if (a != a) return b;
    if (b != b) return a;
    if (a == b) return signbit(a) ? b : a;
    return a < b ? b : a;
}

/* "ocl-amd" flavour of extend.cl:6-27 (see uvrt_oracle.h): fmaf() is exact also without FMA
 * hardware (glibc software fma) */
static inline float dot3_fma(float ax, float ay, float az, float bx, float by, float bz)
{
    return fmaf(az, bz, fmaf(ay, by, ax * bx));
}
/* extend.cl:6-27 as the reference's own build flags compile it for gfx950 (disassembly of
 * oracle/_ref/ref_extend_fast.co): the fused cross / dot of the "ocl-amd" flavour, f = v_rcp_f32(a), and the
 * early returns in the NaN-insensitive forms -cl-fast-relaxed-math licenses: |a| >= 1e-5 continues, 0 <= u && 1 >= u
 * continues, 0 <= v && 1 >= u + v continues */
static inline void intersect_tri_shipped(orc_ray* ray, const orc_tri* tri, uint32_t triID)
{
    const float e1x = tri->v1x - tri->v0x, e1y = tri->v1y - tri->v0y, e1z = tri->v1z - tri->v0z;
    const float e2x = tri->v2x - tri->v0x, e2y = tri->v2y - tri->v0y, e2z = tri->v2z - tri->v0z;
    const float hx = fmaf(ray->diry, e2z, -(ray->dirz * e2y));
    const float hy = fmaf(ray->dirz, e2x, -(ray->dirx * e2z));
    const float hz = fmaf(ray->dirx, e2y, -(ray->diry * e2x));
    const float a = dot3_fma(e1x, e1y, e1z, hx, hy, hz);
    if (!(fabsf(a) >= 0.00001f)) return;
    const float f = hw_rcp(a);
    const float sx = ray->origx - tri->v0x, sy = ray->origy - tri->v0y, sz = ray->origz - tri->v0z;
    const float u = dot3_fma(sx, sy, sz, hx, hy, hz) * f;
    if (!(0.0f <= u && 1.0f >= u)) return;
    const float qx = fmaf(sy, e1z, -(sz * e1y));
    const float qy = fmaf(sz, e1x, -(sx * e1z));
    const float qz = fmaf(sx, e1y, -(sy * e1x));
    const float v = dot3_fma(ray->dirx, ray->diry, ray->dirz, qx, qy, qz) * f;
    if (!(0.0f <= v && 1.0f >= v + u)) return;
    const float t = dot3_fma(e2x, e2y, e2z, qx, qy, qz) * f;
    if (0.0001f < t && t < ray->dist) {
        ray->dist = t;
        ray->triID = triID;
    }
}

static inline void intersect_tri_ocl(orc_ray* ray, const orc_tri* tri, uint32_t triID)
{
    const float e1x = tri->v1x - tri->v0x, e1y = tri->v1y - tri->v0y, e1z = tri->v1z - tri->v0z;
    const float e2x = tri->v2x - tri->v0x, e2y = tri->v2y - tri->v0y, e2z = tri->v2z - tri->v0z;
    const float hx = fmaf(ray->diry, e2z, -(ray->dirz * e2y));
    const float hy = fmaf(ray->dirz, e2x, -(ray->dirx * e2z));
    const float hz = fmaf(ray->dirx, e2y, -(ray->diry * e2x));
    const float a = dot3_fma(e1x, e1y, e1z, hx, hy, hz);
    if (fabsf(a) < 0.00001f) return;
    const float f = 1.0f / a;
    const float sx = ray->origx - tri->v0x, sy = ray->origy - tri->v0y, sz = ray->origz - tri->v0z;
    const float u = f * dot3_fma(sx, sy, sz, hx, hy, hz);
    if ((u < 0) | (u > 1)) return;
    const float qx = fmaf(sy, e1z, -(sz * e1y));
    const float qy = fmaf(sz, e1x, -(sx * e1z));
    const float qz = fmaf(sx, e1y, -(sy * e1x));
    const float v = f * dot3_fma(ray->dirx, ray->diry, ray->dirz, qx, qy, qz);
    if ((v < 0) | (u + v > 1)) return;
    const float t = f * dot3_fma(e2x, e2y, e2z, qx, qy, qz);
    if (t > 0.0001f && t < ray->dist) {
        ray->dist = t;
        ray->triID = triID;
    }
}

/* cl/extend.cl:6-27 */
static inline void intersect_tri(orc_ray* ray, const orc_tri* tri, uint32_t triID)
{
    if (g_flavour == 1) { intersect_tri_ocl(ray, tri, triID); return; }
    if (g_flavour == 2) { intersect_tri_shipped(ray, tri, triID); return; }
    const float e1x = tri->v1x - tri->v0x, e1y = tri->v1y - tri->v0y, e1z = tri->v1z - tri->v0z;
    const float e2x = tri->v2x - tri->v0x, e2y = tri->v2y - tri->v0y, e2z = tri->v2z - tri->v0z;
    /* h = cross(dir, edge2) */
    const float hx = ray->diry * e2z - ray->dirz * e2y;
    const float hy = ray->dirz * e2x - ray->dirx * e2z;
    const float hz = ray->dirx * e2y - ray->diry * e2x;
    const float a = e1x * hx + e1y * hy + e1z * hz;
    if (fabsf(a) < 0.00001f) return;
    const float f = 1.0f / a;
    const float sx = ray->origx - tri->v0x, sy = ray->origy - tri->v0y, sz = ray->origz - tri->v0z;
    const float u = f * (sx * hx + sy * hy + sz * hz);
    if ((u < 0) | (u > 1)) return;
    /* q = cross(s, edge1) */
    const float qx = sy * e1z - sz * e1y;
    const float qy = sz * e1x - sx * e1z;
    const float qz = sx * e1y - sy * e1x;
    const float v = f * (ray->dirx * qx + ray->diry * qy + ray->dirz * qz);
    if ((v < 0) | (u + v > 1)) return;
    const float t = f * (e2x * qx + e2y * qy + e2z * qz);
    if (t > 0.0001f && t < ray->dist) {
        ray->dist = t;
        ray->triID = triID;
    }
}

/* cl/extend.cl:29-38 as the reference's own build flags compile it for gfx950: t = (b - o) * v_rcp_f32(d),
 * v_min_f32 / v_max_f32 / v_max3_f32 / v_min3_f32 in the source's operand order */
static inline float intersect_aabb_shipped(const orc_ray* ray, const orc_node* node)
{
    const float rx = hw_rcp(ray->dirx), ry = hw_rcp(ray->diry), rz = hw_rcp(ray->dirz);
    const float tx1 = (node->minx - ray->origx) * rx, tx2 = (node->maxx - ray->origx) * rx;
    const float ty1 = (node->miny - ray->origy) * ry, ty2 = (node->maxy - ray->origy) * ry;
    const float tz1 = (node->minz - ray->origz) * rz, tz2 = (node->maxz - ray->origz) * rz;
    const float tmin = hw_maxf(hw_maxf(hw_minf(tx1, tx2), hw_minf(ty1, ty2)), hw_minf(tz1, tz2));
    const float tmax = hw_minf(hw_minf(hw_maxf(tx1, tx2), hw_maxf(ty1, ty2)), hw_maxf(tz1, tz2));
    /* the compiled form of extend.cl:37 (no-NaN licence): a miss is tmax < tmin, or tmax >= tmin with !(0 < tmax) or
     * tmin >= dist.  It differs from the source's form only when tmin or tmax is NaN, which takes a direction whose
     * three components are all zero or NaN -- outside the parity domain of this flavour (the product calls that a miss) */
    int miss = tmax < tmin;
    if (tmax >= tmin && (!(0.0f < tmax) || tmin >= ray->dist)) miss = 1;
    return miss ? 1e30f : tmin;
}

/* cl/extend.cl:29-38 */
static inline float intersect_aabb(const orc_ray* ray, const orc_node* node)
{
    if (g_flavour == 2) return intersect_aabb_shipped(ray, node);
    float tx1 = (node->minx - ray->origx) / ray->dirx, tx2 = (node->maxx - ray->origx) / ray->dirx;
    float tmin = cl_minf(tx1, tx2), tmax = cl_maxf(tx1, tx2);
    float ty1 = (node->miny - ray->origy) / ray->diry, ty2 = (node->maxy - ray->origy) / ray->diry;
    tmin = cl_maxf(tmin, cl_minf(ty1, ty2)), tmax = cl_minf(tmax, cl_maxf(ty1, ty2));
    float tz1 = (node->minz - ray->origz) / ray->dirz, tz2 = (node->maxz - ray->origz) / ray->dirz;
    tmin = cl_maxf(tmin, cl_minf(tz1, tz2)), tmax = cl_minf(tmax, cl_maxf(tz1, tz2));
    if (tmax >= tmin && tmin < ray->dist && tmax > 0) return tmin; else return 1e30f;
}

/* cl/extend.cl:40-81 */
static inline void bvh_intersect(orc_ray* ray, const orc_tri* tri, const orc_node* bvhNode,
                                 const uint32_t* triIdx, orc_stats* st)
{
    const orc_node* node = &bvhNode[0];
    const orc_node* stack[32];
    uint32_t stackPtr = 0;
    while (1) {
        st->node_visits++;
        if (st->visit_hist) __atomic_fetch_add(&st->visit_hist[node - bvhNode], 1u, __ATOMIC_RELAXED);
        if (node->triCount > 0) {
            for (uint32_t i = 0; i < (uint32_t)node->triCount; i++) {
                uint32_t triID = triIdx[node->leftFirst + i];
                st->tri_tests++;
                intersect_tri(ray, &tri[triID], triID);
            }
            if (stackPtr == 0) break; else node = stack[--stackPtr];
            continue;
        }
        const orc_node* child1 = &bvhNode[node->leftFirst];
        const orc_node* child2 = &bvhNode[node->leftFirst + 1];
        float dist1 = intersect_aabb(ray, child1);
        float dist2 = intersect_aabb(ray, child2);
        st->aabb_tests += 2;
        if (dist1 > dist2) {
            float d = dist1; dist1 = dist2; dist2 = d;
            const orc_node* c = child1; child1 = child2; child2 = c;
        }
        if (dist1 == 1e30f) {
            if (stackPtr == 0) break; else node = stack[--stackPtr];
        } else {
            node = child1;
            if (dist2 != 1e30f) {
                stack[stackPtr++] = child2;
                if (stackPtr > st->max_stack) st->max_stack = stackPtr;
            }
        }
    }
}

void orc_extend(int32_t* tempPhotonMap, const orc_tri* tris, orc_ray* rays, int64_t n,
                const orc_node* nodes, const uint32_t* triIdx, orc_stats* stats, int nthreads)
{
    orc_stats total;
    memset(&total, 0, sizeof total);
    total.rays = (uint64_t)n;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        orc_stats st;
        memset(&st, 0, sizeof st);
#pragma omp for schedule(dynamic, 4096)
        for (int64_t i = 0; i < n; i++) {
            orc_ray* r = &rays[i];
            bvh_intersect(r, tris, nodes, triIdx, &st);
            if (r->dist != 1e30f) {                                      /* :94-98 */
                st.hits++;
                __atomic_fetch_add(&tempPhotonMap[r->triID], 1, __ATOMIC_RELAXED);
            }
        }
#pragma omp critical
        {
            total.node_visits += st.node_visits;
            total.aabb_tests += st.aabb_tests;
            total.tri_tests += st.tri_tests;
            total.hits += st.hits;
            if (st.max_stack > total.max_stack) total.max_stack = st.max_stack;
        }
    }
    if (stats) *stats = total;
}

/* node visits (traversal steps) of each ray -- analysis helper for the wave-scheduler model in
 * tests/tools/wave_model.py; rays are not modified */
void orc_extend_steps(const orc_tri* tris, const orc_ray* rays, int64_t n, const orc_node* nodes,
                      const uint32_t* triIdx, uint16_t* steps)
{
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = 0; i < n; i++) {
        orc_ray r = rays[i];
        orc_stats st;
        memset(&st, 0, sizeof st);
        bvh_intersect(&r, tris, nodes, triIdx, &st);
        steps[i] = (uint16_t)(st.node_visits > 65535 ? 65535 : st.node_visits);
    }
}

/* visits per node index over n rays -- analysis helper (which tree levels a cache should hold) */
void orc_extend_visit_hist(const orc_tri* tris, const orc_ray* rays, int64_t n, const orc_node* nodes,
                           const uint32_t* triIdx, uint32_t* hist)
{
#pragma omp parallel for schedule(dynamic, 4096)
    for (int64_t i = 0; i < n; i++) {
        orc_ray r = rays[i];
        orc_stats st;
        memset(&st, 0, sizeof st);
        st.visit_hist = hist;
        bvh_intersect(&r, tris, nodes, triIdx, &st);
    }
}

/* Analysis helper (tests/tools/wave_sim.py): the persistent-wave scheduler of the HIP kernel replayed on the CPU with the real
 * arithmetic -- 64 lanes per wave, one traversal step per lane and trip, refill when `refill_min` lanes are idle.
 *   mode 0 = the kernel as it is: a lane at a leaf tests its triangles on a leaf trip (every second trip, or any trip
 *            without a lane at an inner node) and waits otherwise;
 *   mode 1 = the deferred triangle queue: a lane at a leaf appends (lane, leaf) to a per-wave queue,
 *            pops and goes on with its CURRENT (stale) dist; the queue is flushed -- entries tested in queue order, ceil(count /
 *            64) full-width triangle blocks -- when it holds >= queue_flush entries, before a refill, or when no lane stands
 *            at an inner node.
 * Counts trips, triangle-block executions, lane visits, and how many rays end with another (dist, triID) than the
 * reference's traversal gives (mode 1 is not exact by construction: a stale dist adds box visits, and a triangle's computed
 * t can undercut the computed entry distance of a box that the reference culled). */
typedef struct {
    uint64_t rays, trips, inner_trips, leaf_blocks, inner_lane_visits, leaf_lane_tests, wait_lane_trips, idle_lane_trips,
             refills, flushes, queue_entries, differ_tri, differ_dist, ref_inner_visits, ref_tri_tests;
} orc_wavesim;

typedef struct { int64_t ray; const orc_node* cur; const orc_node* stack[32]; uint32_t sp; int active; orc_ray r; } sim_lane;

void orc_wave_sim(const orc_tri* tris, const orc_ray* rays, int64_t n, const orc_node* nodes, const uint32_t* triIdx,
                  int mode, int refill_min, int queue_flush, int rays_per_wave, orc_wavesim* out, int nthreads)
{
    orc_wavesim total;
    memset(&total, 0, sizeof total);
    const int64_t nwaves = (n + rays_per_wave - 1) / rays_per_wave;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel num_threads(nthreads)
    {
        orc_wavesim st;
        memset(&st, 0, sizeof st);
        sim_lane* L = (sim_lane*)malloc(64 * sizeof(sim_lane));
        struct { int lane; const orc_node* leaf; } queue[128];
#pragma omp for schedule(dynamic, 16)
        for (int64_t w = 0; w < nwaves; w++) {
            int64_t next = w * rays_per_wave;
            const int64_t end = (w + 1) * rays_per_wave < n ? (w + 1) * rays_per_wave : n;
            for (int l = 0; l < 64; l++) L[l].active = 0, L[l].cur = NULL;
            int qn = 0, leaf_flag = 1;
            for (;;) {
                int idle = 0, inner = 0, atleaf = 0;
                for (int l = 0; l < 64; l++) {
                    if (!L[l].active || L[l].cur == NULL) idle++;
                    else if (L[l].cur->triCount > 0) atleaf++;
                    else inner++;
                }
                /* flush of the deferred queue */
                const int want_refill = (next < end && idle >= refill_min) || (next >= end && idle == 64);
                if (mode == 1 && qn > 0 && (qn >= queue_flush || want_refill || inner == 0)) {
                    for (int q = 0; q < qn; q++) {
                        sim_lane* o = &L[queue[q].lane];
                        const orc_node* nd = queue[q].leaf;
                        for (uint32_t i = 0; i < (uint32_t)nd->triCount; i++) {
                            const uint32_t id = triIdx[nd->leftFirst + i];
                            intersect_tri(&o->r, &tris[id], id);
                        }
                    }
                    st.flushes++;
                    st.leaf_blocks += (uint64_t)((qn + 63) / 64);
                    st.queue_entries += (uint64_t)qn;
                    qn = 0;
                }
                if (want_refill) {
                    /* finished lanes: compare with the reference's own traversal, then take new rays */
                    for (int l = 0; l < 64; l++) {
                        if (L[l].active && L[l].cur == NULL) {
                            orc_ray ref = rays[L[l].ray];
                            orc_stats rs;
                            memset(&rs, 0, sizeof rs);
                            bvh_intersect(&ref, tris, nodes, triIdx, &rs);
                            st.ref_inner_visits += rs.aabb_tests / 2;
                            st.ref_tri_tests += rs.tri_tests;
                            union { float f; uint32_t u; } a, b;
                            a.f = ref.dist; b.f = L[l].r.dist;
                            if (a.u != b.u) st.differ_dist++;
                            if (ref.triID != L[l].r.triID && !(ref.dist == 1e30f && L[l].r.dist == 1e30f)) st.differ_tri++;
                            L[l].active = 0;
                            st.rays++;
                        }
                    }
                    if (next >= end) break;
                    for (int l = 0; l < 64 && next < end; l++) {
                        if (L[l].active) continue;
                        L[l].active = 1;
                        L[l].ray = next;
                        L[l].r = rays[next++];
                        L[l].cur = &nodes[0];
                        L[l].sp = 0;
                    }
                    st.refills++;
                    continue;
                }
                /* one trip */
                st.trips++;
                if (inner) st.inner_trips++;
                const int leaf_trip = mode == 0 ? (inner == 0 || leaf_flag) : 1;
                leaf_flag = !leaf_flag;
                int tested = 0;
                for (int l = 0; l < 64; l++) {
                    sim_lane* o = &L[l];
                    if (!o->active || o->cur == NULL) { st.idle_lane_trips++; continue; }
                    const orc_node* node = o->cur;
                    if (node->triCount > 0) {
                        if (mode == 0) {
                            if (!leaf_trip) { st.wait_lane_trips++; continue; }
                            for (uint32_t i = 0; i < (uint32_t)node->triCount; i++) {
                                const uint32_t id = triIdx[node->leftFirst + i];
                                intersect_tri(&o->r, &tris[id], id);
                            }
                            tested++;
                        } else {
                            queue[qn].lane = l; queue[qn].leaf = node; qn++;
                        }
                        st.leaf_lane_tests++;
                        o->cur = o->sp ? o->stack[--o->sp] : NULL;
                        continue;
                    }
                    st.inner_lane_visits++;
                    const orc_node* c1 = &nodes[node->leftFirst];
                    const orc_node* c2 = &nodes[node->leftFirst + 1];
                    float d1 = intersect_aabb(&o->r, c1), d2 = intersect_aabb(&o->r, c2);
                    if (d1 > d2) { float d = d1; d1 = d2; d2 = d; const orc_node* c = c1; c1 = c2; c2 = c; }
                    if (d1 == 1e30f) o->cur = o->sp ? o->stack[--o->sp] : NULL;
                    else { o->cur = c1; if (d2 != 1e30f) o->stack[o->sp++] = c2; }
                }
                if (mode == 0 && tested) st.leaf_blocks++;
            }
        }
        free(L);
#pragma omp critical
        {
            uint64_t* a = (uint64_t*)&total; const uint64_t* b = (const uint64_t*)&st;
            for (size_t i = 0; i < sizeof(orc_wavesim) / 8; i++) a[i] += b[i];
        }
    }
    *out = total;
}

/* ------------------------------------------------- accumulate / reset / shade kernels */

void orc_accumulate(double* photonMap, double* maxPhotonMap, int32_t* temp, float timeStep,
                    int32_t T)
{
    for (int32_t i = 0; i < T; i++) {
        photonMap[i] = photonMap[i] + (double)temp[i] * (double)timeStep;   /* :9 */
        double c = (double)temp[i];
        maxPhotonMap[i] = maxPhotonMap[i] < c ? c : maxPhotonMap[i];        /* :11 */
        temp[i] = 0;                                                        /* :13 */
    }
}

void orc_reset(double* photonMap, double* maxPhotonMap, int32_t* temp, orc_tricolor* colorMap,
               int32_t resetColor, int32_t T)
{
    for (int32_t i = 0; i < T; i++) {
        photonMap[i] = 0;
        maxPhotonMap[i] = 0;
        temp[i] = 0;
        if (!resetColor) continue;
        memset(&colorMap[i], 0, sizeof(orc_tricolor));     /* (float)0.0 nine times, :15-25 */
    }
}

void orc_compute_dosage(const double* photonMap, float* dosageMap, const orc_tri* tris,
                        int32_t photonsPerLight, float scaledPower, int32_t T)
{
    for (int32_t i = 0; i < T; i++) {
        const orc_tri* t = &tris[i];
        /* :33-36 */
        float ax = t->v0x - t->v1x, ay = t->v0y - t->v1y, az = t->v0z - t->v1z;
        float bx = t->v0x - t->v2x, by = t->v0y - t->v2y, bz = t->v0z - t->v2z;
        float cx = ay * bz - az * by;
        float cy = az * bx - ax * bz;
        float cz = ax * by - ay * bx;
        float area = sqrtf(cx * cx + cy * cy + cz * cz) / 2.0f;
        /* :39 -- f32*f64 -> f64 ; f32*(int->f32) -> f32 ; f64/f32 -> f64 ; narrowed to f32 */
        float dose = (float)(((double)scaledPower * photonMap[i]) /
                             (double)(area * (float)photonsPerLight));
        dosageMap[i] = dose;
    }
}

/* cl/shade.cl:4-21 */
static void heatmap(float intensity, float rgb[3])
{
    float minDosageColor = 0.5f;
    float upperHalfColor = (float)((double)minDosageColor + (1.0 - (double)minDosageColor) / 2);
    float lowerHalfColor = minDosageColor / 2.0f;
    if (intensity > minDosageColor) {
        if (intensity > upperHalfColor) {
            rgb[0] = 1.0f; rgb[1] = (1.0f - intensity) / (1.0f - upperHalfColor); rgb[2] = 0;
        } else {
            rgb[0] = (intensity - minDosageColor) / (upperHalfColor - minDosageColor);
            rgb[1] = 1.0f; rgb[2] = 0;
        }
    } else {
        if (intensity > lowerHalfColor) {
            rgb[0] = 0; rgb[1] = 1.0f;
            rgb[2] = (minDosageColor - intensity) / (minDosageColor - lowerHalfColor);
        } else {
            rgb[0] = 0; rgb[1] = intensity / lowerHalfColor; rgb[2] = 1.0f;
        }
    }
}

void orc_dosage_to_color(const float* dosageMap, orc_tricolor* colorMap, float minValue,
                         int32_t thresholdView, int32_t T)
{
    for (int32_t i = 0; i < T; i++) {
        float maxValue = minValue * 2;                                    /* :49 */
        float normValue = dosageMap[i] / maxValue;                        /* :51 */
        float c[3];
        if (thresholdView && normValue < 0.5f) {                          /* :56-57 */
            c[0] = 0; c[1] = 0; c[2] = normValue * 2.0f;
        } else {
            heatmap(normValue, c);
        }
        orc_tricolor* tc = &colorMap[i];
        tc->v0x = c[0]; tc->v0y = c[1]; tc->v0z = c[2];
        tc->v1x = c[0]; tc->v1y = c[1]; tc->v1z = c[2];
        tc->v2x = c[0]; tc->v2y = c[1]; tc->v2z = c[2];
    }
}

/* ------------------------------------------------------- floor height: mesh.cpp:100-136 */

float orc_floor_height(const float* yvals, int32_t count)
{
    enum { binCount = 48 };
    float maxVal = 0.0f, minVal = 0.0f;
    int hist[binCount];
    for (int j = 0; j < binCount; ++j) hist[j] = 0;
    for (int32_t i = 0; i < count; ++i)
        if (yvals[i] < minVal) minVal = yvals[i];
    float range = maxVal - minVal;
    for (int32_t i = 0; i < count; ++i) {
        float y = yvals[i];
        for (int j = 0; j < binCount; ++j) {
            /* :120 -- int*float -> float, / int -> float, + float */
            if ((float)j * range / (float)binCount + minVal < y &&
                y < (float)(j + 1) * range / (float)binCount + minVal)
                hist[j]++;
        }
    }
    int maxCount = 0, maxIndex = -1;
    for (int i = 0; i < binCount; ++i)
        if (hist[i] > maxCount) { maxIndex = i; maxCount = hist[i]; }
    return ((float)maxIndex + 0.5f) * range / (float)binCount + minVal;   /* :135 */
}

/* ------------------------------------------------------------------ BVH: bvh.cpp:5-220 */

#define ORC_BINS 8

typedef struct {
    orc_tri* tris;
    int32_t T;
    orc_node* nodes;
    int32_t cap;
    uint32_t* triIdx;
    int32_t extent;           /* highest written node index + 1 */
    int overflow;
    struct { uint32_t nodeIdx; float cmin[3], cmax[3]; } job[64];
    int njobs;
} bvh_ctx;

/* _mm_min_ps(a,b) / _mm_max_ps(a,b) lane semantics */
static inline float sse_min(float a, float b) { return a < b ? a : b; }
static inline float sse_max(float a, float b) { return a > b ? a : b; }

static inline const float* tri_vert(const orc_tri* t, int k) { return &t->v0x + 4 * k; }
static inline const float* tri_centroid(const orc_tri* t) { return &t->cx; }

/* bvh.cpp:181-200 (USE_SSE path): vertex AABB into the node, centroid bounds by reference */
static void update_node_bounds(bvh_ctx* c, uint32_t nodeIdx, float cmin[3], float cmax[3])
{
    orc_node* node = &c->nodes[nodeIdx];
    float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
    float cn[3] = {1e30f, 1e30f, 1e30f}, cx[3] = {-1e30f, -1e30f, -1e30f};
    uint32_t first = (uint32_t)node->leftFirst;
    for (uint32_t i = 0; i < (uint32_t)node->triCount; i++) {
        const orc_tri* t = &c->tris[c->triIdx[first + i]];
        for (int k = 0; k < 3; k++) {
            const float* v = tri_vert(t, k);
            for (int a = 0; a < 3; a++) {
                mn[a] = sse_min(mn[a], v[a]);
                mx[a] = sse_max(mx[a], v[a]);
            }
        }
        const float* ce = tri_centroid(t);
        for (int a = 0; a < 3; a++) {
            cn[a] = sse_min(cn[a], ce[a]);
            cx[a] = sse_max(cx[a], ce[a]);
        }
    }
    node->minx = mn[0]; node->miny = mn[1]; node->minz = mn[2];
    node->maxx = mx[0]; node->maxy = mx[1]; node->maxz = mx[2];
    for (int a = 0; a < 3; a++) { cmin[a] = cn[a]; cmax[a] = cx[a]; }
}

/* bvh.cpp:98-179 (USE_SSE path) */
static float find_best_split(bvh_ctx* c, const orc_node* node, int* axis, int* splitPos,
                             const float cmin[3], const float cmax[3])
{
    float bestCost = 1e30f;
    for (int a = 0; a < 3; a++) {
        float boundsMin = cmin[a], boundsMax = cmax[a];
        if (boundsMin == boundsMax) continue;
        float scale = (float)ORC_BINS / (boundsMax - boundsMin);
        float leftCountArea[ORC_BINS - 1], rightCountArea[ORC_BINS - 1];
        int leftSum = 0, rightSum = 0;
        float bmin[ORC_BINS][3], bmax[ORC_BINS][3];
        uint32_t count[ORC_BINS];
        for (int i = 0; i < ORC_BINS; i++) {
            for (int k = 0; k < 3; k++) { bmin[i][k] = 1e30f; bmax[i][k] = -1e30f; }
            count[i] = 0;
        }
        for (uint32_t i = 0; i < (uint32_t)node->triCount; i++) {
            const orc_tri* t = &c->tris[c->triIdx[node->leftFirst + i]];
            int binIdx = (int)((tri_centroid(t)[a] - boundsMin) * scale);
            if (binIdx > ORC_BINS - 1) binIdx = ORC_BINS - 1;               /* :119 */
            count[binIdx]++;
            for (int k = 0; k < 3; k++) {
                const float* v = tri_vert(t, k);
                for (int d = 0; d < 3; d++) {
                    bmin[binIdx][d] = sse_min(bmin[binIdx][d], v[d]);
                    bmax[binIdx][d] = sse_max(bmax[binIdx][d], v[d]);
                }
            }
        }
        /* :129-143.  NOTE the reference sweeps the right COUNT over bins 7-i but the right
         * BOX over bins 6-i (index BINS-2-i): restated as written, not as intended. */
        float lmin[3] = {1e30f, 1e30f, 1e30f}, rmin[3] = {1e30f, 1e30f, 1e30f};
        float lmax[3] = {-1e30f, -1e30f, -1e30f}, rmax[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = 0; i < ORC_BINS - 1; i++) {
            leftSum += (int)count[i];
            rightSum += (int)count[ORC_BINS - 1 - i];
            float le[3], re[3];
            for (int d = 0; d < 3; d++) {
                lmin[d] = sse_min(lmin[d], bmin[i][d]);
                rmin[d] = sse_min(rmin[d], bmin[ORC_BINS - 2 - i][d]);
                lmax[d] = sse_max(lmax[d], bmax[i][d]);
                rmax[d] = sse_max(rmax[d], bmax[ORC_BINS - 2 - i][d]);
                le[d] = lmax[d] - lmin[d];
                re[d] = rmax[d] - rmin[d];
            }
            leftCountArea[i] = (float)leftSum * (le[0] * le[1] + le[1] * le[2] + le[2] * le[0]);
            rightCountArea[ORC_BINS - 2 - i] =
                (float)rightSum * (re[0] * re[1] + re[1] * re[2] + re[2] * re[0]);
        }
        for (int i = 0; i < ORC_BINS - 1; i++) {                           /* :171-176 */
            const float planeCost = leftCountArea[i] + rightCountArea[i];
            if (planeCost < bestCost) { *axis = a; *splitPos = i + 1; bestCost = planeCost; }
        }
    }
    return bestCost;
}

static void touch(bvh_ctx* c, uint32_t idx)
{
    if ((int32_t)idx + 1 > c->extent) c->extent = (int32_t)idx + 1;
}

/* bvh.cpp:46-96 */
static void subdivide(bvh_ctx* c, uint32_t nodeIdx, uint32_t depth, uint32_t* nodePtr,
                      float cmin[3], float cmax[3])
{
    orc_node* node = &c->nodes[nodeIdx];
    int axis = 0, splitPos = 0;
    float splitCost = find_best_split(c, node, &axis, &splitPos, cmin, cmax);
    /* bvh.h:16-20 CalculateNodeCost */
    float ex = node->maxx - node->minx, ey = node->maxy - node->miny, ez = node->maxz - node->minz;
    float nosplitCost = (ex * ey + ey * ez + ez * ex) * (float)(uint32_t)node->triCount;
    if (splitCost >= nosplitCost) return;
    int i = node->leftFirst;
    int j = i + node->triCount - 1;
    float scale = (float)ORC_BINS / (cmax[axis] - cmin[axis]);
    while (i <= j) {
        int binIdx = (int)((tri_centroid(&c->tris[c->triIdx[i]])[axis] - cmin[axis]) * scale);
        if (binIdx > ORC_BINS - 1) binIdx = ORC_BINS - 1;
        if (binIdx < splitPos) i++;
        else { uint32_t t = c->triIdx[i]; c->triIdx[i] = c->triIdx[j]; c->triIdx[j] = t; j--; }
    }
    int leftCount = i - node->leftFirst;
    if (leftCount == 0 || leftCount == node->triCount) return;
    if ((int32_t)(*nodePtr) + 2 > c->cap) { c->overflow = 1; return; }
    uint32_t leftChildIdx = (*nodePtr)++;
    uint32_t rightChildIdx = (*nodePtr)++;
    touch(c, rightChildIdx);
    c->nodes[leftChildIdx].leftFirst = node->leftFirst;
    c->nodes[leftChildIdx].triCount = leftCount;
    c->nodes[rightChildIdx].leftFirst = i;
    c->nodes[rightChildIdx].triCount = node->triCount - leftCount;
    node->leftFirst = (int32_t)leftChildIdx;
    node->triCount = 0;
    update_node_bounds(c, leftChildIdx, cmin, cmax);
    if (depth == 3) {
        c->job[c->njobs].nodeIdx = leftChildIdx;
        memcpy(c->job[c->njobs].cmin, cmin, sizeof(float) * 3);
        memcpy(c->job[c->njobs].cmax, cmax, sizeof(float) * 3);
        c->njobs++;
    } else subdivide(c, leftChildIdx, depth + 1, nodePtr, cmin, cmax);
    update_node_bounds(c, rightChildIdx, cmin, cmax);
    if (depth == 3) {
        c->job[c->njobs].nodeIdx = rightChildIdx;
        memcpy(c->job[c->njobs].cmin, cmin, sizeof(float) * 3);
        memcpy(c->job[c->njobs].cmax, cmax, sizeof(float) * 3);
        c->njobs++;
    } else subdivide(c, rightChildIdx, depth + 1, nodePtr, cmin, cmax);
}

int32_t orc_bvh_build(orc_tri* tris, int32_t T, orc_node* nodes, int32_t nodes_cap,
                      uint32_t* triIdx)
{
    if (T <= 0 || nodes_cap < 2 * T + 64) return -1;
    bvh_ctx c;
    memset(&c, 0, sizeof c);
    c.tris = tris; c.T = T; c.nodes = nodes; c.cap = nodes_cap; c.triIdx = triIdx;
    uint32_t nodesUsed = 2;                                               /* :16 */
    memset(nodes, 0, sizeof(orc_node) * (size_t)nodes_cap);               /* :17 (whole pool) */
    for (int32_t i = 0; i < T; i++) triIdx[i] = (uint32_t)i;              /* :19 */
    for (int32_t i = 0; i < T; i++) {                                     /* :23 */
        orc_tri* t = &tris[i];
        t->cx = (t->v0x + t->v1x + t->v2x) * 0.3333f;
        t->cy = (t->v0y + t->v1y + t->v2y) * 0.3333f;
        t->cz = (t->v0z + t->v1z + t->v2z) * 0.3333f;
    }
    nodes[0].leftFirst = 0; nodes[0].triCount = T;                        /* :25-26 */
    c.extent = 1;
    float cmin[3], cmax[3];
    update_node_bounds(&c, 0, cmin, cmax);                                /* :28 */
    c.njobs = 0;
    subdivide(&c, 0, 0, &nodesUsed, cmin, cmax);                          /* :31 */
    uint32_t nodePtr[64];                                                 /* :33-36 */
    int N = c.njobs;
    nodePtr[0] = nodesUsed;
    for (int i = 1; i < N; i++)
        nodePtr[i] = nodePtr[i - 1] + (uint32_t)nodes[c.job[i - 1].nodeIdx].triCount * 2;
    for (int i = 0; i < N; i++) {                                         /* :38-42, serial */
        float jmin[3], jmax[3];
        memcpy(jmin, c.job[i].cmin, sizeof jmin);
        memcpy(jmax, c.job[i].cmax, sizeof jmax);
        subdivide(&c, c.job[i].nodeIdx, 99, &nodePtr[i], jmin, jmax);
    }
    if (c.overflow) return -1;
    return c.extent;
}
