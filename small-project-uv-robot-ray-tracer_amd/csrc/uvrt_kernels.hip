// uvrt_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the UV-dose hot path.
//
// Compile with -ffp-contract=off and correctly rounded f32 divide/sqrt (csrc/Makefile): every
// result must equal the reference kernels' strict-IEEE arithmetic bit for bit, so each
// operator below is one rounding in the order the reference source writes it.
//
//   k_generate        cl/generate.cl:8-40       + f64 reciprocals, optional coherence key / rank
//   k_scan_bins, k_scatter                      optional counting sort of rays by coherence key
//   k_extend_persist  cl/extend.cl:6-99         BVH traversal + photon deposit -- THE hot loop
//                                               (persistent waves, in-wave refill; default)
//   k_extend_staged                             same, node records staged through LDS (A/B variant)
//   k_extend                                    v1: one ray per lane, IEEE divisions (A/B variant)
//   k_accumulate      cl/accumulate.cl:4-14
//   k_reset           cl/reset.cl:4-26
//   k_compute_dosage  cl/shade.cl:23-41
//   k_dosage_to_color cl/shade.cl:4-21,43-71
#include "uvrt_device.h"

namespace uvrt {

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

// --------------------------------------------------------------- generate, cl/generate.cl

// Spread the quantised (azimuth, elevation, height) of a photon over one key so that rays
// close in the key are close in space; most-significant bits first, round-robin over the
// three coordinates.  Only the ORDER in which rays are traced depends on it.
__device__ __forceinline__ uint32_t coherence_key(uint32_t qphi, uint32_t qy, uint32_t qo,
                                                  int bphi, int by, int bo)
{
    uint32_t key = 0;
    while (bphi > 0 || by > 0 || bo > 0) {
        if (bphi > 0) { --bphi; key = (key << 1) | ((qphi >> bphi) & 1u); }
        if (by > 0) { --by; key = (key << 1) | ((qy >> by) & 1u); }
        if (bo > 0) { --bo; key = (key << 1) | ((qo >> bo) & 1u); }
    }
    return key;
}

__global__ __launch_bounds__(256) void k_generate(GenParams p)
{
    if (blockIdx.x >= p.ray_blocks) {   // extra workgroups: extend v6's per-launch node-pair records
        const int j = (int)(blockIdx.x - p.ray_blocks) * 256 + threadIdx.x;
        if (j < p.prep_npairs) prepare_record6(p.prep_pairs, p.prep_recs, p.lx, p.lz, p.prep_npairs, j, p.prep_perm);
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const int64_t gid = p.first_gid + i;
    const int threadID = (int)gid;                       // generate.cl:11 (int threadID)
    const uint32_t SEED = (gid == 0 || p.seed_mode == 1) ? p.seed_prev : p.seed_next;

    // generate.cl:13 -- f32 adds in source order, then float -> uint through int64
    float acc = (float)(threadID * 17 + 1);
    acc = acc + p.lx * 13.0f;
    acc = acc + p.ly * 7.0f;
    acc = acc + p.lz * 11.0f;
    acc = acc + (float)(SEED >> 15);
    uint32_t seed = wang_hash((p.seed_mode == 1 && acc < 0.0f) ? 0u : (uint32_t)(int64_t)acc);

    const float r0 = random_float(seed);
    const float origy = p.ly + r0 * p.light_length;      // :16
    const float diry = random_float(seed) * 2.0f - 1.0f; // :22
    const double dirxzlength = sqrt(1.0 - (double)diry * (double)diry);   // :23

    double x = (double)(random_float(seed) * 2.0f - 1.0f);                // :25
    double y = (double)(random_float(seed) * 2.0f - 1.0f);
    while (x * x + y * y > 1.0) {                                         // :26-28
        x = (double)(random_float(seed) * 2.0f - 1.0f);
        y = (double)(random_float(seed) * 2.0f - 1.0f);
    }
    const double s = dirxzlength / sqrt(x * x + y * y);                   // :29
    const float dirx = (float)(x * s);
    const float dirz = (float)(y * s);
    p.rays[i] = make_float4(dirx, diry, dirz, origy);                     // :31-37
    if (p.recip) {   // RN64(1/dir): the slab test's exact-division shortcut (see slab<>())
        p.recip[i] = 1.0 / (double)dirx;
        p.recip[p.recip_stride + i] = 1.0 / (double)diry;
        p.recip[2 * p.recip_stride + i] = 1.0 / (double)dirz;
    }

    if (p.keyrank) {
        // azimuth as a diamond angle in [0,4): monotone in the true angle, one division
        const float fx = (float)x, fy = (float)y;
        const float ax = fabsf(fx), ay = fabsf(fy);
        float t = ay / (ax + ay + 1e-30f);
        float ang = fx >= 0.0f ? (fy >= 0.0f ? t : 4.0f - t) : (fy >= 0.0f ? 2.0f - t : 2.0f + t);
        const uint32_t nphi = 1u << p.bits_phi, ny = 1u << p.bits_y, no = 1u << p.bits_o;
        uint32_t qphi = min((uint32_t)(ang * (0.25f * (float)nphi)), nphi - 1);
        uint32_t qy = min((uint32_t)((diry + 1.0f) * (0.5f * (float)ny)), ny - 1);
        uint32_t qo = min((uint32_t)(r0 * (float)no), no - 1);
        const uint32_t key = coherence_key(qphi, qy, qo, p.bits_phi, p.bits_y, p.bits_o);
        const uint32_t rank = atomicAdd(&p.hist[key], 1u);
        p.keyrank[i] = make_uint2(key, rank);
    }
}

// Exclusive prefix sum over the key histogram (one 1024-thread workgroup; nbins <= 2^20), and
// re-zero the histogram for the next launch.
__global__ __launch_bounds__(1024) void k_scan_bins(uint32_t* hist, uint32_t* bin_start,
                                                    int32_t nbins)
{
    __shared__ uint32_t s_part[1024];
    const int tid = threadIdx.x;
    const int per = (nbins + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, nbins);
    uint32_t sum = 0;
    for (int b = lo; b < hi; ++b) sum += hist[b];
    s_part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = tid >= off ? s_part[tid - off] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_part[tid] - sum;
    for (int b = lo; b < hi; ++b) {
        const uint32_t c = hist[b];
        bin_start[b] = run;
        hist[b] = 0;
        run += c;
    }
}

__global__ __launch_bounds__(256) void k_scatter(const float4* __restrict__ rays,
                                                 const uint2* __restrict__ keyrank,
                                                 const uint32_t* __restrict__ bin_start,
                                                 float4* __restrict__ sorted,
                                                 uint32_t* __restrict__ order,
                                                 double* __restrict__ recip, int64_t recip_stride,
                                                 int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint2 kr = keyrank[i];
    const uint32_t pos = bin_start[kr.x] + kr.y;
    const float4 r = rays[i];
    sorted[pos] = r;
    order[pos] = (uint32_t)i;
    if (recip) {
        recip[pos] = 1.0 / (double)r.x;
        recip[recip_stride + pos] = 1.0 / (double)r.y;
        recip[2 * recip_stride + pos] = 1.0 / (double)r.z;
    }
}

// RN64(1/dir) for rays that were generated without them (extend v6 does not need them)
__global__ __launch_bounds__(256) void k_fill_recip(const float4* __restrict__ rays, double* __restrict__ recip,
                                                    int64_t recip_stride, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 r = rays[i];
    recip[i] = 1.0 / (double)r.x;
    recip[recip_stride + i] = 1.0 / (double)r.y;
    recip[2 * recip_stride + i] = 1.0 / (double)r.z;
}

// ------------------------------------------------------------------- extend, cl/extend.cl

// OpenCL min/max on floats: min(x,y) = y<x ? y : x, max(x,y) = x<y ? y : x.  They differ from
// v_min_f32/v_max_f32 only when an operand is NaN, and a slab distance is NaN only for 0/0,
// i.e. when a direction component is exactly zero.  NANSAFE=false uses the hardware min/max
// and is selected per wave when no lane has a zero direction component.
template <bool NANSAFE>
__device__ __forceinline__ float cl_min(float x, float y)
{
    if (NANSAFE) return y < x ? y : x;
    return __builtin_fminf(x, y);
}
template <bool NANSAFE>
__device__ __forceinline__ float cl_max(float x, float y)
{
    if (NANSAFE) return x < y ? y : x;
    return __builtin_fmaxf(x, y);
}

struct RayRegs {
    float ox, oy, oz;
    float dx, dy, dz;
    float dist;
    uint32_t triID;
};

// extend.cl:29-38 -- six IEEE divisions by the direction, as written
template <bool NANSAFE>
__device__ __forceinline__ float intersect_aabb(const RayRegs& r, float mnx, float mny, float mnz,
                                                float mxx, float mxy, float mxz)
{
    const float tx1 = (mnx - r.ox) / r.dx, tx2 = (mxx - r.ox) / r.dx;
    float tmin = cl_min<NANSAFE>(tx1, tx2), tmax = cl_max<NANSAFE>(tx1, tx2);
    const float ty1 = (mny - r.oy) / r.dy, ty2 = (mxy - r.oy) / r.dy;
    tmin = cl_max<NANSAFE>(tmin, cl_min<NANSAFE>(ty1, ty2));
    tmax = cl_min<NANSAFE>(tmax, cl_max<NANSAFE>(ty1, ty2));
    const float tz1 = (mnz - r.oz) / r.dz, tz2 = (mxz - r.oz) / r.dz;
    tmin = cl_max<NANSAFE>(tmin, cl_min<NANSAFE>(tz1, tz2));
    tmax = cl_min<NANSAFE>(tmax, cl_max<NANSAFE>(tz1, tz2));
    if (tmax >= tmin && tmin < r.dist && tmax > 0) return tmin;
    return 1e30f;
}

// extend.cl:6-27 with edge1/edge2 read from the leaf record
__device__ __forceinline__ void intersect_tri(RayRegs& r, const LeafTri* __restrict__ t)
{
    const float4 v0 = t->v0_id, e1 = t->e1, e2 = t->e2;
    const float hx = r.dy * e2.z - r.dz * e2.y;
    const float hy = r.dz * e2.x - r.dx * e2.z;
    const float hz = r.dx * e2.y - r.dy * e2.x;
    const float a = e1.x * hx + e1.y * hy + e1.z * hz;
    if (fabsf(a) < 0.00001f) return;
    const float f = 1.0f / a;
    const float sx = r.ox - v0.x, sy = r.oy - v0.y, sz = r.oz - v0.z;
    const float u = f * (sx * hx + sy * hy + sz * hz);
    if ((u < 0) | (u > 1)) return;
    const float qx = sy * e1.z - sz * e1.y;
    const float qy = sz * e1.x - sx * e1.z;
    const float qz = sx * e1.y - sy * e1.x;
    const float v = f * (r.dx * qx + r.dy * qy + r.dz * qz);
    if ((v < 0) | (u + v > 1)) return;
    const float tt = f * (e2.x * qx + e2.y * qy + e2.z * qz);
    if (tt > 0.0001f && tt < r.dist) {
        r.dist = tt;
        r.triID = __float_as_uint(v0.w);
    }
}

constexpr int LDS_STACK = 16;   // stack entries kept in LDS per lane; 16 more live in scratch
constexpr int MAX_STACK = 32;   // extend.cl:43

// extend.cl:40-81.  The node visit order, every AABB / triangle test and every comparison are
// the reference's; only the record layout differs.  Per-lane traversal stack: entry e of lane
// l at s_stack[e][l] (bank = lane, conflict free at any mix of depths).
template <bool NANSAFE>
__device__ __forceinline__ void bvh_intersect(RayRegs& r, const SceneDev& sc,
                                              uint32_t (*s_stack)[256], uint32_t* error_flag)
{
    const int tid = threadIdx.x;
    uint32_t ovf[MAX_STACK - LDS_STACK];
    int sp = 0;
    uint32_t cur = sc.root_ref;

#define UVRT_POP()                                                                     \
    do {                                                                               \
        if (sp == 0) cur = REF_DONE;                                                   \
        else {                                                                         \
            --sp;                                                                      \
            if (sp < LDS_STACK) { cur = s_stack[sp][tid]; asm volatile("" : "+v"(cur)); } \
            else cur = ovf[sp - LDS_STACK];                                            \
        }                                                                              \
    } while (0)

    while (cur != REF_DONE) {
        while (cur < REF_LEAF_BIT) {                       // inner node: test both children
            const PairRec* pr = sc.pairs + cur;
            const float4 a = pr->c0min_ref0, b = pr->c0max_ref1, c = pr->c1min, d = pr->c1max;
            float dist1 = intersect_aabb<NANSAFE>(r, a.x, a.y, a.z, b.x, b.y, b.z);
            float dist2 = intersect_aabb<NANSAFE>(r, c.x, c.y, c.z, d.x, d.y, d.z);
            uint32_t ref1 = __float_as_uint(a.w), ref2 = __float_as_uint(b.w);
            if (dist1 > dist2) {                           // extend.cl:61-65
                const float td = dist1; dist1 = dist2; dist2 = td;
                const uint32_t tr = ref1; ref1 = ref2; ref2 = tr;
            }
            if (dist1 == 1e30f) {                          // :66-69
                UVRT_POP();
            } else {                                       // :70-76
                cur = ref1;
                if (dist2 != 1e30f) {
                    if (sp < LDS_STACK) s_stack[sp][tid] = ref2;
                    else if (sp < MAX_STACK) ovf[sp - LDS_STACK] = ref2;
                    else *error_flag = 1u;
                    if (sp < MAX_STACK) ++sp;
                }
            }
        }
        if (cur != REF_DONE) {                             // leaf: extend.cl:48-55
            const uint32_t first = cur & REF_FIRST_MASK;
            uint32_t count = (cur >> REF_COUNT_SHIFT) & 15u;
            if (count == 15u) count = sc.leaf_count[first];
            for (uint32_t i = 0; i < count; ++i) intersect_tri(r, sc.ltris + first + i);
            UVRT_POP();
        }
    }
#undef UVRT_POP
}

__global__ __launch_bounds__(256) void k_extend(ExtendParams p)
{
    __shared__ uint32_t s_stack[LDS_STACK][256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const float4 rec = p.rays[i];
    RayRegs r;
    r.ox = p.ox; r.oy = rec.w; r.oz = p.oz;
    r.dx = rec.x; r.dy = rec.y; r.dz = rec.z;
    r.dist = 1e30f;                                        // generate.cl:34-35
    r.triID = 0;
    const bool zero_dir = (r.dx == 0.0f) | (r.dy == 0.0f) | (r.dz == 0.0f);
    if (__any(zero_dir)) bvh_intersect<true>(r, p.scene, s_stack, p.error_flag);
    else bvh_intersect<false>(r, p.scene, s_stack, p.error_flag);

    if (p.hits) {
        const uint32_t li = p.order ? p.order[i] : (uint32_t)i;
        p.hits[li] = make_uint2(__float_as_uint(r.dist), r.triID);
    }
    if (r.dist != 1e30f)                                     // extend.cl:94-98
        atomicAdd(&p.counts[(int64_t)(blockIdx.x % (unsigned)p.count_replicas) * p.count_stride + r.triID], 1);
}


// ----------------------------------------------------------------------------------------
// extend v2: persistent wavefronts, one traversal step per loop trip, idle lanes refilled from
// a global ray counter, and the slab divisions taken through a precomputed f64 reciprocal.
//
// slab<EXACT=false>:  (float)((double)(b - o) * r)  with  r = RN64(1 / (double)d)
// equals the IEEE binary32 quotient  RN32((b - o) / d)  bit for bit.  Proof sketch (DESIGN.md
// "Exact division by reciprocal"): let a = RN32(b - o).  (i) The f64 product is (a/d)(1+e),
// |e| <= 2^-52.  (ii) For binary32 a, d the exact quotient a/d is never closer than 2^-49
// (relative) to a rounding boundary of binary32 -- a midpoint M*2^k with M odd in (2^24,2^25),
// or the overflow threshold -- unless it is far into the subnormal range: a - M*2^k*d is a
// non-zero integer multiple of 2^(k + exponent(d)), and |d| < 2^24 ulps.  Hence the product and
// the quotient lie on the same side of every boundary and round to the same float.  (iii) The
// quotient is normal-or-zero whenever |d| <= 1 and a is zero or |a| >= 2^-100; lanes or scenes
// outside these conditions (zero / >1 direction component, tiny origin or bound) take
// slab<true>, the reference's own division, as do NaN-producing rays (0/0).  +-inf from d = 0
// never reaches this path.
template <bool EXACT>
__device__ __forceinline__ float slab(float b, float o, float d, double r)
{
    if (EXACT) return (b - o) / d;
    return (float)((double)(b - o) * r);
}

struct RayState {
    float ox, oy, oz;
    float dx, dy, dz;
    double rx, ry, rz;
    float dist;
    uint32_t triID;
};

template <bool EXACT>
__device__ __forceinline__ float intersect_aabb2(const RayState& r, float mnx, float mny, float mnz,
                                                 float mxx, float mxy, float mxz)
{
    const float tx1 = slab<EXACT>(mnx, r.ox, r.dx, r.rx), tx2 = slab<EXACT>(mxx, r.ox, r.dx, r.rx);
    float tmin = cl_min<EXACT>(tx1, tx2), tmax = cl_max<EXACT>(tx1, tx2);
    const float ty1 = slab<EXACT>(mny, r.oy, r.dy, r.ry), ty2 = slab<EXACT>(mxy, r.oy, r.dy, r.ry);
    tmin = cl_max<EXACT>(tmin, cl_min<EXACT>(ty1, ty2));
    tmax = cl_min<EXACT>(tmax, cl_max<EXACT>(ty1, ty2));
    const float tz1 = slab<EXACT>(mnz, r.oz, r.dz, r.rz), tz2 = slab<EXACT>(mxz, r.oz, r.dz, r.rz);
    tmin = cl_max<EXACT>(tmin, cl_min<EXACT>(tz1, tz2));
    tmax = cl_min<EXACT>(tmax, cl_max<EXACT>(tz1, tz2));
    if (tmax >= tmin && tmin < r.dist && tmax > 0) return tmin;
    return 1e30f;
}

__device__ __forceinline__ void intersect_tri2(RayState& r, const LeafTri* __restrict__ t)
{
    const float4 v0 = t->v0_id, e1 = t->e1, e2 = t->e2;
    const float hx = r.dy * e2.z - r.dz * e2.y;
    const float hy = r.dz * e2.x - r.dx * e2.z;
    const float hz = r.dx * e2.y - r.dy * e2.x;
    const float a = e1.x * hx + e1.y * hy + e1.z * hz;
    if (fabsf(a) < 0.00001f) return;
    const float f = 1.0f / a;
    const float sx = r.ox - v0.x, sy = r.oy - v0.y, sz = r.oz - v0.z;
    const float u = f * (sx * hx + sy * hy + sz * hz);
    if ((u < 0) | (u > 1)) return;
    const float qx = sy * e1.z - sz * e1.y;
    const float qy = sz * e1.x - sx * e1.z;
    const float qz = sx * e1.y - sy * e1.x;
    const float v = f * (r.dx * qx + r.dy * qy + r.dz * qz);
    if ((v < 0) | (u + v > 1)) return;
    const float tt = f * (e2.x * qx + e2.y * qy + e2.z * qz);
    if (tt > 0.0001f && tt < r.dist) {
        r.dist = tt;
        r.triID = __float_as_uint(v0.w);
    }
}

// One traversal step of one lane (extend.cl:44-80): an inner node (test both children, order
// them, descend / push / pop) or a leaf (test its triangles, pop).
// Top-of-tree cache: the first `top_pairs` node-pair records (breadth-first numbering: the upper
// levels of the tree) are copied into LDS by every workgroup.  Part p of record r sits in 16-byte
// slot (p + (r >> 2)) & 3 of its 64-byte block, which spreads the lanes of a ds_read_b128 over all
// sixteen 4-bank windows instead of four.
constexpr int TOP_MAX_PAIRS = 127;        // 7 complete levels; 8 KB of LDS with the padding record
constexpr int PSTACK = 8;                 // LDS stack entries per lane in the persistent kernel

__device__ __forceinline__ int top_slot(uint32_t r, int p) { return (int)((p + (r >> 2)) & 3u); }

// Moeller-Trumbore on a leaf record already in registers (extend.cl:6-27)
// "ocl-amd" flavour of the triangle test (include/uvrt.h uvrt_set_flavour): cross and dot as the
// fused forms of ROCm's OpenCL library, everything else as extend.cl writes it
__device__ __forceinline__ float dot3_fma(float ax, float ay, float az, float bx, float by, float bz)
{
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}
__device__ __forceinline__ void intersect_tri_ocl(RayState& r, const float4 v0, const float4 e1, const float4 e2)
{
    const float hx = __builtin_fmaf(r.dy, e2.z, -(r.dz * e2.y));
    const float hy = __builtin_fmaf(r.dz, e2.x, -(r.dx * e2.z));
    const float hz = __builtin_fmaf(r.dx, e2.y, -(r.dy * e2.x));
    const float a = dot3_fma(e1.x, e1.y, e1.z, hx, hy, hz);
    if (fabsf(a) < 0.00001f) return;
    const float f = 1.0f / a;
    const float sx = r.ox - v0.x, sy = r.oy - v0.y, sz = r.oz - v0.z;
    const float u = f * dot3_fma(sx, sy, sz, hx, hy, hz);
    if ((u < 0) | (u > 1)) return;
    const float qx = __builtin_fmaf(sy, e1.z, -(sz * e1.y));
    const float qy = __builtin_fmaf(sz, e1.x, -(sx * e1.z));
    const float qz = __builtin_fmaf(sx, e1.y, -(sy * e1.x));
    const float v = f * dot3_fma(r.dx, r.dy, r.dz, qx, qy, qz);
    if ((v < 0) | (u + v > 1)) return;
    const float tt = f * dot3_fma(e2.x, e2.y, e2.z, qx, qy, qz);
    if (tt > 0.0001f && tt < r.dist) {
        r.dist = tt;
        r.triID = __float_as_uint(v0.w);
    }
}

__device__ __forceinline__ void intersect_tri_regs(RayState& r, const float4 v0, const float4 e1, const float4 e2)
{
    const float hx = r.dy * e2.z - r.dz * e2.y;
    const float hy = r.dz * e2.x - r.dx * e2.z;
    const float hz = r.dx * e2.y - r.dy * e2.x;
    const float a = e1.x * hx + e1.y * hy + e1.z * hz;
    if (fabsf(a) < 0.00001f) return;
    const float f = 1.0f / a;
    const float sx = r.ox - v0.x, sy = r.oy - v0.y, sz = r.oz - v0.z;
    const float u = f * (sx * hx + sy * hy + sz * hz);
    if ((u < 0) | (u > 1)) return;
    const float qx = sy * e1.z - sz * e1.y;
    const float qy = sz * e1.x - sx * e1.z;
    const float qz = sx * e1.y - sy * e1.x;
    const float v = f * (r.dx * qx + r.dy * qy + r.dz * qz);
    if ((v < 0) | (u + v > 1)) return;
    const float tt = f * (e2.x * qx + e2.y * qy + e2.z * qz);
    if (tt > 0.0001f && tt < r.dist) {
        r.dist = tt;
        r.triID = __float_as_uint(v0.w);
    }
}

// One traversal step of one lane (extend.cl:44-80).  The texture-data unit spends ~29 cycles on
// every vector-load wave-instruction whatever its active lanes (profiles/: TD busy = 29 x VMEM
// instructions, 85 % of the kernel), so inner-node lanes and leaf lanes share ONE set of four
// load instructions per trip: the record address is a per-lane select between the pair array and
// the leaf-triangle array, and the fourth 16 bytes are only used by inner lanes.
template <bool EXACT, int NSTACK, bool TOP, bool OCL>
__device__ __forceinline__ void traversal_step(RayState& r, uint32_t& cur, int& sp, uint32_t* ovf,
                                               const SceneDev& sc, uint32_t (*s_stack)[256],
                                               const float4* s_top, uint32_t top_pairs,
                                               uint32_t* error_flag)
{
    const int tid = threadIdx.x;
    const bool is_inner = cur < REF_LEAF_BIT;
    const bool is_leaf = !is_inner && cur != REF_DONE;
    const uint32_t first = cur & REF_FIRST_MASK;
    // byte offset of the record from the pair array's base; the leaf array is addressed relative
    // to the same base so the select is one 64-bit value per lane
    const char* base = (const char*)sc.pairs;
    const int64_t off = is_inner ? (int64_t)cur * (int64_t)sizeof(PairRec)
                                 : ((const char*)sc.ltris - base) + (int64_t)first * (int64_t)sizeof(LeafTri);
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f w0, w1, w2, w3;   // deliberately not initialised: written by the loads below, read only
                          // by the lanes that executed them (zero-filling costs 16 VALU per trip)
    uint32_t spec_top = REF_DONE;
    if (is_inner | is_leaf) {
        if (sp > 0 && sp <= NSTACK) spec_top = s_stack[sp - 1][tid];
        // exactly four 16-byte loads for inner and leaf lanes alike (hipcc would re-split them
        // into six odd-sized ones); leaf lanes over-read 16 bytes, the leaf array is padded for it
        const char* recp = base + off;
        asm volatile("global_load_dwordx4 %0, %4, off\n\t"
                     "global_load_dwordx4 %1, %4, off offset:16\n\t"
                     "global_load_dwordx4 %2, %4, off offset:32\n\t"
                     "global_load_dwordx4 %3, %4, off offset:48\n\t"
                     "s_waitcnt vmcnt(0)"
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                     : "v"(recp)
                     : "memory");
    }
    bool pop = false;
    if (is_inner) {
        float dist1 = intersect_aabb2<EXACT>(r, w0.x, w0.y, w0.z, w1.x, w1.y, w1.z);
        float dist2 = intersect_aabb2<EXACT>(r, w2.x, w2.y, w2.z, w3.x, w3.y, w3.z);
        uint32_t ref1 = __float_as_uint(w0.w), ref2 = __float_as_uint(w1.w);
        if (dist1 > dist2) {
            const float td = dist1; dist1 = dist2; dist2 = td;
            const uint32_t tr = ref1; ref1 = ref2; ref2 = tr;
        }
        if (dist1 == 1e30f) pop = true;
        else {
            cur = ref1;
            if (dist2 != 1e30f) {
                if (sp < NSTACK) s_stack[sp][tid] = ref2;
                else if (sp < MAX_STACK) ovf[sp - NSTACK] = ref2;
                else *error_flag = 1u;
                if (sp < MAX_STACK) ++sp;
            }
        }
    } else if (is_leaf) {
        uint32_t count = (cur >> REF_COUNT_SHIFT) & 15u;
        if (count == 15u) count = sc.leaf_count[first];
        const float4 t0 = make_float4(w0.x, w0.y, w0.z, w0.w), t1 = make_float4(w1.x, w1.y, w1.z, w1.w),
                     t2 = make_float4(w2.x, w2.y, w2.z, w2.w);
        if (!OCL) {
            intersect_tri_regs(r, t0, t1, t2);
            for (uint32_t i = 1; i < count; ++i) intersect_tri2(r, sc.ltris + first + i);
        } else {
            intersect_tri_ocl(r, t0, t1, t2);
            for (uint32_t i = 1; i < count; ++i) {
                const LeafTri* lt = sc.ltris + first + i;
                intersect_tri_ocl(r, lt->v0_id, lt->e1, lt->e2);
            }
        }
        pop = true;
    }
    if (pop) {
        if (sp == 0) cur = REF_DONE;
        else {
            --sp;
            if (sp < NSTACK) cur = spec_top;
            else cur = ovf[sp - NSTACK];
        }
    }
}

// Persistent wavefronts over statically owned rays.  The 64-ray batches of the launch are dealt
// round-robin to the waves of the grid; lanes that finish a ray take the next unclaimed ray of
// their own wave's sequence (a wave-uniform cursor: no atomics, no inter-wave traffic), REFILL_MIN
// idle lanes at a time, so the 64 lanes stay busy although ray lengths differ by an order of
// magnitude (mean 32 steps, max ~200).
template <int REFILL_MIN, bool TOP, bool OCL = false>
__global__ __launch_bounds__(256, 8) void k_extend_persist(ExtendParams p)
{
    __shared__ uint32_t s_stack[PSTACK][256];                       // 8 KB
    __shared__ float4 s_top[TOP ? (TOP_MAX_PAIRS + 1) * 4 : 4];     // 8 KB
    // stack entries 8..31 of this thread (0.02 % of pushes on the test room) live in global memory
    uint32_t* const ovf = p.ovf_stack + ((size_t)blockIdx.x * 256 + threadIdx.x) * (MAX_STACK - PSTACK);
    const uint32_t top_pairs = TOP ? p.top_pairs : 0u;
    if (TOP) {
        const float4* src = (const float4*)p.scene.pairs;
        for (uint32_t i = threadIdx.x; i < top_pairs * 4u; i += 256u) {
            const uint32_t rec = i >> 2;
            s_top[rec * 4u + (uint32_t)top_slot(rec, (int)(i & 3u))] = src[i];
        }
        __syncthreads();
    }
    RayState r;
    r.ox = p.ox; r.oz = p.oz;
    r.oy = 0.f; r.dx = r.dy = r.dz = 1.f; r.rx = r.ry = r.rz = 1.0; r.dist = 1e30f; r.triID = 0;
    uint32_t cur = REF_DONE;   // this lane holds no ray
    uint32_t slot = 0;         // trace slot of the ray held
    int sp = 0;
    bool live = false;         // holds a ray whose result has not been deposited yet
    bool special = false;      // this lane's ray needs the EXACT path
    int32_t* const my_counts = p.counts + (int64_t)(blockIdx.x % (unsigned)p.count_replicas) * p.count_stride;

    // Wave w traces the 64-ray batches w, w + W, w + 2W, ... (W = waves in the grid): `cursor`
    // counts rays of that private sequence, sequence element v is trace slot
    // ((v / 64) * W + w) * 64 + v % 64.  Dealing batches round-robin keeps waves balanced when
    // rays are ordered by direction (neighbouring batches have similar traversal lengths).
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t W = gridDim.x * 4u;
    uint32_t cursor = 0;
    const uint32_t chunk_end = p.chunk;   // sequence length; slots >= n are skipped
    const uint32_t n32 = (uint32_t)p.n;

    for (;;) {
        const bool idle = cur == REF_DONE;
        const unsigned long long idle_mask = __ballot(idle);
        const int nidle = __popcll(idle_mask);
        if (cursor < chunk_end && nidle >= REFILL_MIN) {
            if (idle) {
                // results of the rays these lanes finished since the last refill: ONE atomic
                // instruction per refill instead of one per loop trip (vector-memory instructions
                // are the scarce resource: ~29 TD cycles each whatever the active lanes)
                if (live) {
                    live = false;
                    if (p.hits) {
                        const uint32_t li = p.order ? p.order[slot] : slot;
                        p.hits[li] = make_uint2(__float_as_uint(r.dist), r.triID);
                    }
                    if (r.dist != 1e30f) atomicAdd(&my_counts[r.triID], 1);   // extend.cl:94-98
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                      __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const uint32_t v = cursor + rank;
                const uint32_t my = ((v >> 6) * W + wave) * 64u + (v & 63u);
                if (v < chunk_end && my < n32) {
                    const float4 rec = p.rays[my];
                    r.dx = rec.x; r.dy = rec.y; r.dz = rec.z; r.oy = rec.w;
                    r.rx = p.recip[my];
                    r.ry = p.recip[p.recip_stride + my];
                    r.rz = p.recip[2 * p.recip_stride + my];
                    r.dist = 1e30f;                        // generate.cl:34-35
                    r.triID = 0;
                    slot = my;
                    sp = 0;
                    cur = p.scene.root_ref;
                    live = true;
                    const float ay = fabsf(r.oy);
                    special = r.dx == 0.0f || r.dy == 0.0f || r.dz == 0.0f ||
                              !(fabsf(r.dx) <= 1.0f) || !(fabsf(r.dy) <= 1.0f) || !(fabsf(r.dz) <= 1.0f) ||
                              (ay != 0.0f && ay < 7.888609e-31f) || p.force_exact != 0;
                }
            }
            cursor += (uint32_t)nidle;
        }
        const bool active = cur != REF_DONE;
        if (!__any(active)) {
            if (cursor >= chunk_end) break;
            continue;
        }
        if (__any(active & special))
            traversal_step<true, PSTACK, TOP, OCL>(r, cur, sp, ovf, p.scene, s_stack, s_top, top_pairs, p.error_flag);
        else
            traversal_step<false, PSTACK, TOP, OCL>(r, cur, sp, ovf, p.scene, s_stack, s_top, top_pairs, p.error_flag);

    }
    // the wave's sequence is exhausted and every lane is idle: deposit what is still pending
    if (live) {
        if (p.hits) {
            const uint32_t li = p.order ? p.order[slot] : slot;
            p.hits[li] = make_uint2(__float_as_uint(r.dist), r.triID);
        }
        if (r.dist != 1e30f) atomicAdd(&my_counts[r.triID], 1);       // extend.cl:94-98
    }
}

// ----------------------------------------------------------------------------------------
// extend v3: as k_extend_persist, but node-pair records are STAGED THROUGH LDS per wavefront.
//
// PMC evidence (profiles/r01_v3_extend_pmc_summary.txt): with one lane fetching its own 64-byte
// record as four dwordx4 loads, the vector L1 takes 243 M accesses per launch (4 per record) and
// is ~72 % busy; waves sit in s_waitcnt 58 % of the time.  Here the four lanes of a quad fetch
// ONE record per load instruction (the quad reads its 64 contiguous bytes: one L1 access), four
// instructions cover the quad's four records, and the 4x4 transpose happens in LDS.  The loads
// are LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write.  L1 accesses per record:
// 4 -> 1.
//
// LDS per wave: 4 regions x 1 KiB; instruction k fills region k lane-linearly (an LDS-DMA
// destination is wave-uniform base + 16*lane), so the record of lane l = 4q + k lies in region k
// at byte 64q.  The bank-conflict swizzle goes on the SOURCE address: lane j of the quad fetches
// logical part j^k, i.e. logical part p of that record sits in 16-byte slot p^k.  Lanes 4q..4q+3
// (k = 0..3, same 64-dword bank window, regions 1 KiB apart) then read four distinct slots in
// each ds_read_b128 -- conflict-free.
constexpr int STG_STACK = 8;   // LDS stack entries per lane in the staged kernel (8 KB / block)

template <bool EXACT>
__device__ __forceinline__ void staged_step(RayState& r, uint32_t& cur, int& sp, uint32_t* ovf,
                                            const SceneDev& sc, uint32_t (*s_stack)[256],
                                            const float4* my_rec, bool have_rec, uint32_t* error_flag)
{
    const int tid = threadIdx.x;
    bool pop = false;
    if (have_rec) {                                        // inner node, record staged in LDS
        const int k = tid & 3;
        const float4 a = my_rec[0 ^ k], b = my_rec[1 ^ k], c = my_rec[2 ^ k], d = my_rec[3 ^ k];
        float dist1 = intersect_aabb2<EXACT>(r, a.x, a.y, a.z, b.x, b.y, b.z);
        float dist2 = intersect_aabb2<EXACT>(r, c.x, c.y, c.z, d.x, d.y, d.z);
        uint32_t ref1 = __float_as_uint(a.w), ref2 = __float_as_uint(b.w);
        if (dist1 > dist2) {
            const float td = dist1; dist1 = dist2; dist2 = td;
            const uint32_t tr = ref1; ref1 = ref2; ref2 = tr;
        }
        if (dist1 == 1e30f) pop = true;
        else {
            cur = ref1;
            if (dist2 != 1e30f) {
                if (sp < STG_STACK) s_stack[sp][tid] = ref2;
                else if (sp < MAX_STACK) ovf[sp - STG_STACK] = ref2;
                else *error_flag = 1u;
                if (sp < MAX_STACK) ++sp;
            }
        }
    } else if (cur != REF_DONE) {                          // leaf (cur has bit 31 set)
        const uint32_t first = cur & REF_FIRST_MASK;
        uint32_t count = (cur >> REF_COUNT_SHIFT) & 15u;
        if (count == 15u) count = sc.leaf_count[first];
        for (uint32_t i = 0; i < count; ++i) intersect_tri2(r, sc.ltris + first + i);
        pop = true;
    }
    if (pop) {
        if (sp == 0) cur = REF_DONE;
        else {
            --sp;
            if (sp < STG_STACK) { cur = s_stack[sp][tid]; asm volatile("" : "+v"(cur)); }
            else cur = ovf[sp - STG_STACK];
        }
    }
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_void_t;

template <int REFILL_MIN>
__global__ __launch_bounds__(256, 6) void k_extend_staged(ExtendParams p)
{
    __shared__ uint32_t s_stack[STG_STACK][256];
    __shared__ float4 s_stage[4][4][64];                   // [wave][region k][lane]
    uint32_t* const ovf = p.ovf_stack + ((size_t)blockIdx.x * 256 + threadIdx.x) * (MAX_STACK - STG_STACK);
    RayState r;
    r.ox = p.ox; r.oz = p.oz;
    r.oy = 0.f; r.dx = r.dy = r.dz = 1.f; r.rx = r.ry = r.rz = 1.0; r.dist = 1e30f; r.triID = 0;
    uint32_t cur = REF_DONE;
    uint32_t slot = 0;
    int sp = 0;
    bool live = false, special = false;
    int32_t* const my_counts = p.counts + (int64_t)(blockIdx.x % (unsigned)p.count_replicas) * p.count_stride;

    const int lane = threadIdx.x & 63;
    const int j = lane & 3;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float4 (*const stage)[64] = s_stage[wv];
    // my own record: region (lane & 3), 64-byte block (lane >> 2)
    const float4* const my_rec = &stage[lane & 3][(lane >> 2) * 4];
    const uint32_t wave = blockIdx.x * 4u + (uint32_t)wv;
    const uint32_t W = gridDim.x * 4u;
    uint32_t cursor = 0;
    const uint32_t chunk_end = p.chunk;
    const uint32_t n32 = (uint32_t)p.n;
    const char* const pairs_bytes = (const char*)p.scene.pairs;

    for (;;) {
        const bool idle = cur == REF_DONE;
        const unsigned long long idle_mask = __ballot(idle);
        const int nidle = __popcll(idle_mask);
        if (cursor < chunk_end && nidle >= REFILL_MIN) {
            if (idle) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                      __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const uint32_t v = cursor + rank;
                const uint32_t my = ((v >> 6) * W + wave) * 64u + (v & 63u);
                if (v < chunk_end && my < n32) {
                    const float4 rec = p.rays[my];
                    r.dx = rec.x; r.dy = rec.y; r.dz = rec.z; r.oy = rec.w;
                    r.rx = p.recip[my];
                    r.ry = p.recip[p.recip_stride + my];
                    r.rz = p.recip[2 * p.recip_stride + my];
                    r.dist = 1e30f;
                    r.triID = 0;
                    slot = my;
                    sp = 0;
                    cur = p.scene.root_ref;
                    live = true;
                    const float ay = fabsf(r.oy);
                    special = r.dx == 0.0f || r.dy == 0.0f || r.dz == 0.0f ||
                              !(fabsf(r.dx) <= 1.0f) || !(fabsf(r.dy) <= 1.0f) || !(fabsf(r.dz) <= 1.0f) ||
                              (ay != 0.0f && ay < 7.888609e-31f) || p.force_exact != 0;
                }
            }
            cursor += (uint32_t)nidle;
        }
        const bool active = cur != REF_DONE;
        if (!__any(active)) {
            if (cursor >= chunk_end) break;
            continue;
        }
        // ---- cooperative fetch: quad q loads the records of its own four lanes ----
        const bool inner = cur < REF_LEAF_BIT;
        if (__any(inner)) {
            // record index of quad lane k, broadcast within the quad (DPP quad_perm, no LDS)
            const uint32_t c0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)cur, 0x00, 0xf, 0xf, true);
            const uint32_t c1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)cur, 0x55, 0xf, 0xf, true);
            const uint32_t c2 = (uint32_t)__builtin_amdgcn_mov_dpp((int)cur, 0xaa, 0xf, 0xf, true);
            const uint32_t c3 = (uint32_t)__builtin_amdgcn_mov_dpp((int)cur, 0xff, 0xf, 0xf, true);
            if (c0 < REF_LEAF_BIT)
                __builtin_amdgcn_global_load_lds((global_void_t*)(pairs_bytes + (size_t)c0 * 64 + ((j ^ 0) * 16)),
                                                 (lds_void_t*)&stage[0][0], 16, 0, 0);
            if (c1 < REF_LEAF_BIT)
                __builtin_amdgcn_global_load_lds((global_void_t*)(pairs_bytes + (size_t)c1 * 64 + ((j ^ 1) * 16)),
                                                 (lds_void_t*)&stage[1][0], 16, 0, 0);
            if (c2 < REF_LEAF_BIT)
                __builtin_amdgcn_global_load_lds((global_void_t*)(pairs_bytes + (size_t)c2 * 64 + ((j ^ 2) * 16)),
                                                 (lds_void_t*)&stage[2][0], 16, 0, 0);
            if (c3 < REF_LEAF_BIT)
                __builtin_amdgcn_global_load_lds((global_void_t*)(pairs_bytes + (size_t)c3 * 64 + ((j ^ 3) * 16)),
                                                 (lds_void_t*)&stage[3][0], 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA landed; same-wave reads follow
        }
        if (__any(active & special))
            staged_step<true>(r, cur, sp, ovf, p.scene, s_stack, my_rec, inner, p.error_flag);
        else
            staged_step<false>(r, cur, sp, ovf, p.scene, s_stack, my_rec, inner, p.error_flag);

        if (live && cur == REF_DONE) {
            live = false;
            if (p.hits) {
                const uint32_t li = p.order ? p.order[slot] : slot;
                p.hits[li] = make_uint2(__float_as_uint(r.dist), r.triID);
            }
            if (r.dist != 1e30f) atomicAdd(&my_counts[r.triID], 1);
        }
    }
}

// ----------------------------------------------------------- per-triangle kernels (O(T))

// accumulate.cl:4-14; tempPhotonMap[i] is the (exact, integer) sum of the deposit replicas.
// Four lanes share a triangle: each sums (and clears) a quarter of the replicas, two DPP adds
// combine them, lane 0 of the quad applies the reference's update.
__global__ __launch_bounds__(256) void k_accumulate(double* __restrict__ photon_map,
                                                    double* __restrict__ max_map,
                                                    int32_t* __restrict__ counts, int32_t replicas,
                                                    int64_t stride, float time_step, int32_t T)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int i = g >> 2, part = g & 3;
    int32_t total = 0;
    if (i < T) {
        for (int r = part; r < replicas; r += 4) {
            total += counts[r * stride + i];
            counts[r * stride + i] = 0;
        }
    }
    total += __builtin_amdgcn_mov_dpp(total, 0xb1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    total += __builtin_amdgcn_mov_dpp(total, 0x4e, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    if (i >= T || part != 0) return;
    const double c = (double)total;
    photon_map[i] = photon_map[i] + c * (double)time_step;
    const double m = max_map[i];
    max_map[i] = m < c ? c : m;
}

// tempPhotonMap in the reference's single-array form: replica 0 = sum, the others zero
__global__ __launch_bounds__(256) void k_fold_counts(int32_t* __restrict__ counts, int32_t replicas,
                                                     int64_t stride, int32_t T)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    int32_t total = counts[i];
    for (int r = 1; r < replicas; ++r) {
        total += counts[r * stride + i];
        counts[r * stride + i] = 0;
    }
    counts[i] = total;
}

// reset.cl:4-26
__global__ __launch_bounds__(256) void k_reset(double* __restrict__ photon_map,
                                               double* __restrict__ max_map,
                                               int32_t* __restrict__ counts, int32_t replicas,
                                               int64_t stride, float* __restrict__ color,
                                               int32_t reset_color, int32_t T)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    photon_map[i] = 0;
    max_map[i] = 0;
    for (int r = 0; r < replicas; ++r) counts[r * stride + i] = 0;
    if (!reset_color) return;
    for (int k = 0; k < 9; ++k) color[(int64_t)i * 9 + k] = 0.0f;
}

// Scene preparation: leaf-ordered triangle records and the per-triangle area of
// shade.cl:33-36 (area = length(cross(v0-v1, v0-v2)) / 2.0f), computed once per scene.
__global__ __launch_bounds__(256) void k_prepare_scene(const float4* __restrict__ tris64,
                                                       const uint32_t* __restrict__ tri_idx,
                                                       LeafTri* __restrict__ ltris,
                                                       float* __restrict__ area, int32_t T)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    {
        const uint32_t id = tri_idx[i];
        const float4 v0 = tris64[(int64_t)id * 4 + 0], v1 = tris64[(int64_t)id * 4 + 1],
                     v2 = tris64[(int64_t)id * 4 + 2];
        LeafTri t;
        t.v0_id = make_float4(v0.x, v0.y, v0.z, __uint_as_float(id));
        t.e1 = make_float4(v1.x - v0.x, v1.y - v0.y, v1.z - v0.z, 0.0f);
        t.e2 = make_float4(v2.x - v0.x, v2.y - v0.y, v2.z - v0.z, 0.0f);
        ltris[i] = t;
    }
    {
        const float4 v0 = tris64[(int64_t)i * 4 + 0], v1 = tris64[(int64_t)i * 4 + 1],
                     v2 = tris64[(int64_t)i * 4 + 2];
        const float ax = v0.x - v1.x, ay = v0.y - v1.y, az = v0.z - v1.z;
        const float bx = v0.x - v2.x, by = v0.y - v2.y, bz = v0.z - v2.z;
        const float cx = ay * bz - az * by;
        const float cy = az * bx - ax * bz;
        const float cz = ax * by - ay * bx;
        area[i] = sqrtf(cx * cx + cy * cy + cz * cz) / 2.0f;
    }
}

// shade.cl:23-41: dose = (scaledPower * photonMap) / (area * photonsPerLight)
//   f32*f64 -> f64 ; f32*(int->f32) -> f32 ; f64/f32 -> f64 ; narrowed to f32
__global__ __launch_bounds__(256) void k_compute_dosage(const double* __restrict__ map,
                                                        float* __restrict__ dosage,
                                                        const float* __restrict__ area,
                                                        int32_t photons_per_light,
                                                        float scaled_power, int32_t T)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    const double num = (double)scaled_power * map[i];
    const float den = area[i] * (float)photons_per_light;
    dosage[i] = (float)(num / (double)den);
}

// shade.cl:4-21
__device__ __forceinline__ void heatmap(float intensity, float& r, float& g, float& b)
{
    const float minDosageColor = 0.5f;
    const float upperHalfColor = 0.75f;   // 0.5f + (1.0 - 0.5f) / 2, exact
    const float lowerHalfColor = 0.25f;   // 0.5f / 2.0f, exact
    if (intensity > minDosageColor) {
        if (intensity > upperHalfColor) {
            r = 1.0f; g = (1.0f - intensity) / (1.0f - upperHalfColor); b = 0.0f;
        } else {
            r = (intensity - minDosageColor) / (upperHalfColor - minDosageColor);
            g = 1.0f; b = 0.0f;
        }
    } else {
        if (intensity > lowerHalfColor) {
            r = 0.0f; g = 1.0f;
            b = (minDosageColor - intensity) / (minDosageColor - lowerHalfColor);
        } else {
            r = 0.0f; g = intensity / lowerHalfColor; b = 1.0f;
        }
    }
}

// shade.cl:43-71
__global__ __launch_bounds__(256) void k_dosage_to_color(const float* __restrict__ dosage,
                                                         float* __restrict__ color,
                                                         float min_value, int32_t threshold_view,
                                                         int32_t T)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    const float maxValue = min_value * 2;
    const float norm = dosage[i] / maxValue;
    float r, g, b;
    if (threshold_view && norm < 0.5f) { r = 0.0f; g = 0.0f; b = norm * 2.0f; }
    else heatmap(norm, r, g, b);
    float* c = color + (int64_t)i * 9;
    c[0] = r; c[1] = g; c[2] = b;
    c[3] = r; c[4] = g; c[5] = b;
    c[6] = r; c[7] = g; c[8] = b;
}

// computeDosage + dosageToColor in one pass (RayTracer::Shade always runs them back to back,
// raytracer.cpp:93-120); same operations, the dose still goes through its f32 store
__global__ __launch_bounds__(256) void k_shade(const double* __restrict__ map, float* __restrict__ dosage,
                                               const float* __restrict__ area, float* __restrict__ color,
                                               int32_t photons_per_light, float scaled_power,
                                               float min_value, int32_t threshold_view, int32_t T)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    const double num = (double)scaled_power * map[i];                   // shade.cl:39
    const float den = area[i] * (float)photons_per_light;
    const float dose = (float)(num / (double)den);
    dosage[i] = dose;
    const float maxValue = min_value * 2;                               // shade.cl:47-50
    const float norm = dose / maxValue;
    float r, g, b;
    if (threshold_view && norm < 0.5f) { r = 0.0f; g = 0.0f; b = norm * 2.0f; }
    else heatmap(norm, r, g, b);
    float* c = color + (int64_t)i * 9;
    c[0] = r; c[1] = g; c[2] = b;
    c[3] = r; c[4] = g; c[5] = b;
    c[6] = r; c[7] = g; c[8] = b;
}

// Test hook: the reference's 32-byte Ray records (cl/tools.cl:8-14) in gid order.
__global__ __launch_bounds__(256) void k_export_rays(const float4* __restrict__ rays,
                                                     const uint2* __restrict__ hits,
                                                     float4* __restrict__ out, float ox,
                                                     float oz, int64_t first, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const float4 rec = rays[first + i];
    uint2 h = make_uint2(__float_as_uint(1e30f), 0u);
    if (hits) h = hits[first + i];
    out[i * 2 + 0] = make_float4(rec.x, rec.y, rec.z, ox);
    out[i * 2 + 1] = make_float4(rec.w, oz, __uint_as_float(h.x), __uint_as_float(h.y));
}

// ------------------------------------------------------------------------ launch wrappers

static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

void launch_generate(const GenParams& p0, hipStream_t s)
{
    GenParams p = p0;
    p.ray_blocks = p.n > 0 ? blocks_for(p.n, 256) : 0u;
    const unsigned prep_blocks = p.prep_recs ? blocks_for(p.prep_npairs, 256) : 0u;
    if (p.ray_blocks + prep_blocks == 0) return;
    hipLaunchKernelGGL(k_generate, dim3(p.ray_blocks + prep_blocks), dim3(256), 0, s, p);
}

void launch_fill_recip(const float4* rays, double* recip, int64_t recip_stride, int64_t n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fill_recip, dim3(blocks_for(n, 256)), dim3(256), 0, s, rays, recip, recip_stride, n);
}

void launch_scan_bins(uint32_t* hist, uint32_t* bin_start, int32_t nbins, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan_bins, dim3(1), dim3(1024), 0, s, hist, bin_start, nbins);
}

void launch_scatter(const float4* rays, const uint2* keyrank, const uint32_t* bin_start,
                    float4* sorted, uint32_t* order, double* recip_sorted, int64_t recip_stride,
                    int64_t n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n, 256)), dim3(256), 0, s, rays, keyrank,
                       bin_start, sorted, order, recip_sorted, recip_stride, n);
}

// variant = threshold code + 10 * grid code.  Threshold code: 0 default (refill at 16 idle
// lanes), 1 = v1 kernel (one ray per lane, IEEE divisions, no refill), 2/3/5/6 = refill when
// >= 1/8/32/64 lanes are idle, 4 = default + the top-of-tree LDS cache (measured: no gain).
// Grid code: workgroups per CU = 8 (0), 4 (1), 6 (2), 2 (3), 16 (4).
bool launch_extend(const ExtendParams& p0, int variant, hipStream_t s)
{
    if (p0.n <= 0) return true;
    ExtendParams p = p0;
    const int tcode = variant % 10, gcode = (variant / 10) % 10;
    if (tcode == 1) {
        hipLaunchKernelGGL(k_extend, dim3(blocks_for(p.n, 256)), dim3(256), 0, s, p);
        return true;
    }
    static const unsigned per_cu[5] = {8, 4, 6, 2, 16};
    const unsigned cus = p.num_cus > 0 ? (unsigned)p.num_cus : 256u;
    unsigned grid = cus * per_cu[gcode < 5 ? gcode : 0];
    const unsigned need = blocks_for(p.n, 256);
    if (need < grid) grid = need;
    const uint64_t waves = (uint64_t)grid * 4;
    p.chunk = (uint32_t)((((uint64_t)p.n + waves - 1) / waves + 63) / 64 * 64);   // whole batches of 64
    if (tcode >= 7) {   // staged kernels: 6 workgroups per CU unless a grid code says otherwise
        if (gcode == 0) { grid = cus * 6u; if (need < grid) grid = need; }
        if ((uint64_t)grid * 256 * (MAX_STACK - STG_STACK) > p.ovf_capacity) return false;
        const uint64_t w2 = (uint64_t)grid * 4;
        p.chunk = (uint32_t)((((uint64_t)p.n + w2 - 1) / w2 + 63) / 64 * 64);
        switch (tcode) {
            case 8: hipLaunchKernelGGL(k_extend_staged<8>, dim3(grid), dim3(256), 0, s, p); break;
            case 9: hipLaunchKernelGGL(k_extend_staged<32>, dim3(grid), dim3(256), 0, s, p); break;
            default: hipLaunchKernelGGL(k_extend_staged<16>, dim3(grid), dim3(256), 0, s, p); break;
        }
        return true;
    }
    if ((uint64_t)grid * 256 * (MAX_STACK - PSTACK) > p.ovf_capacity) return false;
    if (p.flavour != 0) {   // "ocl-amd" triangle arithmetic: its own instantiation of the default kernel
        hipLaunchKernelGGL((k_extend_persist<16, false, true>), dim3(grid), dim3(256), 0, s, p);
        return true;
    }
    switch (tcode) {
        case 2: hipLaunchKernelGGL((k_extend_persist<1, false>), dim3(grid), dim3(256), 0, s, p); break;
        case 3: hipLaunchKernelGGL((k_extend_persist<8, false>), dim3(grid), dim3(256), 0, s, p); break;
        case 4: hipLaunchKernelGGL((k_extend_persist<16, true>), dim3(grid), dim3(256), 0, s, p); break;   // + top-of-tree LDS cache
        case 5: hipLaunchKernelGGL((k_extend_persist<32, false>), dim3(grid), dim3(256), 0, s, p); break;
        case 6: hipLaunchKernelGGL((k_extend_persist<64, false>), dim3(grid), dim3(256), 0, s, p); break;
        default: hipLaunchKernelGGL((k_extend_persist<16, false>), dim3(grid), dim3(256), 0, s, p); break;
    }
    return true;
}

void launch_accumulate(double* photon_map, double* max_map, int32_t* counts, int32_t replicas,
                       int64_t stride, float time_step, int32_t T, hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(k_accumulate, dim3(blocks_for((int64_t)T * 4, 256)), dim3(256), 0, s, photon_map,
                       max_map, counts, replicas, stride, time_step, T);
}

void launch_fold_counts(int32_t* counts, int32_t replicas, int64_t stride, int32_t T, hipStream_t s)
{
    if (T <= 0 || replicas <= 1) return;
    hipLaunchKernelGGL(k_fold_counts, dim3(blocks_for(T, 256)), dim3(256), 0, s, counts, replicas,
                       stride, T);
}

void launch_reset(double* photon_map, double* max_map, int32_t* counts, int32_t replicas,
                  int64_t stride, float* color, int32_t reset_color, int32_t T, hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(k_reset, dim3(blocks_for(T, 256)), dim3(256), 0, s, photon_map, max_map,
                       counts, replicas, stride, color, reset_color, T);
}

void launch_compute_dosage(const double* map, float* dosage, const float* area,
                           int32_t photons_per_light, float scaled_power, int32_t T,
                           hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(k_compute_dosage, dim3(blocks_for(T, 256)), dim3(256), 0, s, map, dosage,
                       area, photons_per_light, scaled_power, T);
}

void launch_shade(const double* map, float* dosage, const float* area, float* color, int32_t photons_per_light,
                  float scaled_power, float min_value, int32_t threshold_view, int32_t T, hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(k_shade, dim3(blocks_for(T, 256)), dim3(256), 0, s, map, dosage, area, color,
                       photons_per_light, scaled_power, min_value, threshold_view, T);
}

void launch_dosage_to_color(const float* dosage, float* color, float min_value,
                            int32_t threshold_view, int32_t T, hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(k_dosage_to_color, dim3(blocks_for(T, 256)), dim3(256), 0, s, dosage,
                       color, min_value, threshold_view, T);
}

void launch_prepare_scene(const float4* tris64, const uint32_t* tri_idx, LeafTri* ltris,
                          float* area, int32_t T, hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(k_prepare_scene, dim3(blocks_for(T, 256)), dim3(256), 0, s, tris64,
                       tri_idx, ltris, area, T);
}

void launch_export_rays(const float4* rays, const uint2* hits, void* out32, float ox, float oz,
                        int64_t first, int64_t count, hipStream_t s)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_export_rays, dim3(blocks_for(count, 256)), dim3(256), 0, s, rays, hits,
                       (float4*)out32, ox, oz, first, count);
}

}  // namespace uvrt
