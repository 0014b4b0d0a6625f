// uvrt_traverse.h -- device helpers shared by the traversal kernels (uvrt_extend6.hip: the reference's
// BVH2 order; uvrt_extend4.hip: the opt-in 4-wide collapse): exact slab distances, box and triangle tests,
// the per-lane ray state, the stack-overflow pointer.  See uvrt_extend6.hip's header for the arithmetic.
#pragma once
#include "uvrt_device.h"

namespace uvrt {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int MAXS6 = 32;                 // extend.cl:43
constexpr int PS6 = 8;                    // LDS stack entries per lane.  7.5 % of the trips of the test room see a deeper
                                          // stack and take the general step (1.6 % with 9 rows, 0.35 % with 10:
                                          // tests/tools/trip_stats.sh), but more rows gain nothing: 9 rows with 123 cached
                                          // records (still eight workgroups per CU) measure the same, 10 rows lose 5 % --
                                          // launch pipelining lives on the eighth workgroup slot of a CU
constexpr uint32_t TOP6_STRIDE = 64;      // bytes per cached record.  (80 with 16 bytes of padding spread the lanes of a
                                          // ds_read_b128 over more banks, but LDS reads cost a trip nothing measurable and 48
                                          // more records do: +2 %, profiles/r02/r02_experiments.txt)

// The six slab distances of one child box (extend.cl:31-37), correctly rounded:
//   t = a / d  as  q0 = a * y;  r = fma(-d, q0, a);  q = fma(r, y, q0),   y = RN32(1/d)
// for the three (min, max) numerator pairs x, z (already a = b - o) and y (raw bounds: the ray's
// origin y is subtracted first).  px/py/pz = {d, y} per axis, po = {origin y, .}; every step is one
// packed instruction with the broadcast of d, y or o done by op_sel.  One asm block so that the
// three chains are interleaved by hand and need exactly three temporary register pairs.
__device__ __forceinline__ void slabs6(v2f& x, v2f& y, v2f& z, v2f px, v2f py, v2f pz, v2f po)
{
    v2f tx, ty, tz;
    asm("v_pk_add_f32 %[y], %[y], %[po] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[tx], %[x], %[px] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %[tz], %[z], %[pz] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %[ty], %[y], %[py] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_fma_f32 %[x], %[px], %[tx], %[x] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %[z], %[pz], %[tz], %[z] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %[y], %[py], %[ty], %[y] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %[x], %[x], %[px], %[tx] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %[z], %[z], %[pz], %[tz] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_fma_f32 %[y], %[y], %[py], %[ty] op_sel:[0,1,0] op_sel_hi:[1,1,1]"
        : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z), [tx] "=&v"(tx), [ty] "=&v"(ty), [tz] "=&v"(tz)
        : [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [po] "v"(po));
}

// The same six distances in the "shipped flags" flavour (uvrt_set_flavour 2): t = (b - o) * v_rcp_f32(d), with the
// reciprocal in the high half of px / py / pz -- what the reference's own build flags make of extend.cl:31-35
__device__ __forceinline__ void slabs6s(v2f& x, v2f& y, v2f& z, v2f px, v2f py, v2f pz, v2f po)
{
    asm("v_pk_add_f32 %[y], %[y], %[po] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_mul_f32 %[x], %[x], %[px] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %[z], %[z], %[pz] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
        "v_pk_mul_f32 %[y], %[y], %[py] op_sel:[0,1] op_sel_hi:[1,1]"
        : [x] "+v"(x), [y] "+v"(y), [z] "+v"(z)
        : [px] "v"(px), [py] "v"(py), [pz] "v"(pz), [po] "v"(po));
}

// extend.cl:29-38 from the three (t at min, t at max) pairs: entry distance and hit flag.  No operand
// is NaN on this path, so v_min/v_max equal OpenCL's y<x?y:x / x<y?y:x.
__device__ __forceinline__ bool box_fast(v2f tx, v2f ty, v2f tz, float dist, float& tmin)
{
    float nx, fx, ny, fy, nz, fz, tmax;
    asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(tx.x), "v"(tx.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(tx.x), "v"(tx.y));
    asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ty.x), "v"(ty.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ty.x), "v"(ty.y));
    asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(tz.x), "v"(tz.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(tz.x), "v"(tz.y));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(nx), "v"(ny), "v"(nz));   // max(max(nx, ny), nz)
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(fx), "v"(fy), "v"(fz));   // min(min(fx, fy), fz)
    return (tmax >= tmin) & (tmin < dist) & (tmax > 0);
}

// extend.cl:29-38 for BOTH child boxes in one block (no hazard padding between statements): entry and exit
// distances from the (t at min, t at max) pairs.  No operand is NaN on this path.
__device__ __forceinline__ void box2_fast(v2f tx0, v2f ty0, v2f tz0, v2f tx1, v2f ty1, v2f tz1, float& tmin0, float& tmax0,
                                          float& tmin1, float& tmax1)
{
    float a, b, c;
    asm("v_min_f32 %[a], %[x0l], %[x0h]\n\t"
        "v_min_f32 %[b], %[y0l], %[y0h]\n\t"
        "v_min_f32 %[c], %[z0l], %[z0h]\n\t"
        "v_max3_f32 %[n0], %[a], %[b], %[c]\n\t"
        "v_max_f32 %[a], %[x0l], %[x0h]\n\t"
        "v_max_f32 %[b], %[y0l], %[y0h]\n\t"
        "v_max_f32 %[c], %[z0l], %[z0h]\n\t"
        "v_min3_f32 %[f0], %[a], %[b], %[c]\n\t"
        "v_min_f32 %[a], %[x1l], %[x1h]\n\t"
        "v_min_f32 %[b], %[y1l], %[y1h]\n\t"
        "v_min_f32 %[c], %[z1l], %[z1h]\n\t"
        "v_max3_f32 %[n1], %[a], %[b], %[c]\n\t"
        "v_max_f32 %[a], %[x1l], %[x1h]\n\t"
        "v_max_f32 %[b], %[y1l], %[y1h]\n\t"
        "v_max_f32 %[c], %[z1l], %[z1h]\n\t"
        "v_min3_f32 %[f1], %[a], %[b], %[c]"
        : [n0] "=&v"(tmin0), [f0] "=&v"(tmax0), [n1] "=&v"(tmin1), [f1] "=&v"(tmax1), [a] "=&v"(a), [b] "=&v"(b), [c] "=&v"(c)
        : [x0l] "v"(tx0.x), [x0h] "v"(tx0.y), [y0l] "v"(ty0.x), [y0h] "v"(ty0.y), [z0l] "v"(tz0.x), [z0h] "v"(tz0.y),
          [x1l] "v"(tx1.x), [x1h] "v"(tx1.y), [y1l] "v"(ty1.x), [y1h] "v"(ty1.y), [z1l] "v"(tz1.x), [z1h] "v"(tz1.y));
}

// the reference's own form: IEEE divisions, OpenCL min/max as selects (NaN operands: 0/0)
__device__ __forceinline__ bool box_exact(float ax1, float ax2, float ay1, float ay2, float az1, float az2,
                                          float dx, float dy, float dz, float dist, float& tmin_out)
{
    const float tx1 = ax1 / dx, tx2 = ax2 / dx;
    float tmin = tx2 < tx1 ? tx2 : tx1, tmax = tx1 < tx2 ? tx2 : tx1;
    const float ty1 = ay1 / dy, ty2 = ay2 / dy;
    const float mny = ty2 < ty1 ? ty2 : ty1, mxy = ty1 < ty2 ? ty2 : ty1;
    tmin = tmin < mny ? mny : tmin;
    tmax = mxy < tmax ? mxy : tmax;
    const float tz1 = az1 / dz, tz2 = az2 / dz;
    const float mnz = tz2 < tz1 ? tz2 : tz1, mxz = tz1 < tz2 ? tz2 : tz1;
    tmin = tmin < mnz ? mnz : tmin;
    tmax = mxz < tmax ? mxz : tmax;
    tmin_out = tmin;
    return tmax >= tmin && tmin < dist && tmax > 0;
}

// RN32(1 / a) for 2^-64 <= |a| < 2^64 (file header)
__device__ __forceinline__ float rcp_exact(float a)
{
    float y0;
    asm("v_rcp_f32 %0, %1" : "=v"(y0) : "v"(a));
    const float e = __builtin_fmaf(-a, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}

// v_rcp_f32 as it is (about 1 ulp): the reciprocal of the "shipped flags" flavour
__device__ __forceinline__ float rcp_raw(float a)
{
    float y;
    asm("v_rcp_f32 %0, %1" : "=v"(y) : "v"(a));
    return y;
}

// extend.cl:6-27 on a leaf record (v0, e1 = v1 - v0, e2 = v2 - v0, id in v0.w).
// FL = the arithmetic flavour (include/uvrt.h uvrt_set_flavour):
//   0  strict: every operator one rounding, source order;
//   1  "ocl-amd": cross() and dot() in the fused forms ROCm's OpenCL device library gives the reference's
//      extend.cl on gfx950 (read off the disassembly of that kernel as built for gfx950):
//      cross(a, b).x = fma(a.y, b.z, -(a.z * b.y)), dot(a, b) = fma(a.z, b.z, fma(a.y, b.y, a.x * b.x));
//      everything else as extend.cl writes it;
//   2  "shipped flags": what the reference's OWN build options (-cl-fast-relaxed-math -cl-mad-enable,
//      template/template.cpp:1192) make of extend.cl on gfx950 (read off the disassembly of that build, which the tests run live beside this): the
//      fused forms of flavour 1, f = v_rcp_f32(a) without refinement, and the early returns in the forms the
//      no-NaN licence gives them (|a| >= 1e-5, 0 <= u, 1 >= u, 0 <= v, 1 >= v + u continue).
template <int FL>
__device__ __forceinline__ float cross6(float ay, float bz, float az, float by)
{
    return FL != 0 ? __builtin_fmaf(ay, bz, -(az * by)) : ay * bz - az * by;
}
template <int FL>
__device__ __forceinline__ float dot6(float ax, float ay, float az, float bx, float by, float bz)
{
    return FL != 0 ? __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx)) : ax * bx + ay * by + az * bz;
}
template <int FL>
__device__ __forceinline__ void tri6(float ox, float oy, float oz, float dx, float dy, float dz, float& dist,
                                     uint32_t& triID, const float4 v0, const float4 e1, const float4 e2,
                                     bool exact)
{
    const float hx = cross6<FL>(dy, e2.z, dz, e2.y);
    const float hy = cross6<FL>(dz, e2.x, dx, e2.z);
    const float hz = cross6<FL>(dx, e2.y, dy, e2.x);
    const float a = dot6<FL>(e1.x, e1.y, e1.z, hx, hy, hz);
    if (FL == 2 ? !(fabsf(a) >= 0.00001f) : fabsf(a) < 0.00001f) return;
    float f;
    if (FL == 2) f = rcp_raw(a);
    else if (exact) f = 1.0f / a;         // wave-uniform
    else f = rcp_exact(a);
    const float sx = ox - v0.x, sy = oy - v0.y, sz = oz - v0.z;
    const float u = f * dot6<FL>(sx, sy, sz, hx, hy, hz);
    if (FL == 2 ? !((0.0f <= u) & (1.0f >= u)) : ((u < 0) | (u > 1))) return;
    const float qx = cross6<FL>(sy, e1.z, sz, e1.y);
    const float qy = cross6<FL>(sz, e1.x, sx, e1.z);
    const float qz = cross6<FL>(sx, e1.y, sy, e1.x);
    const float v = f * dot6<FL>(dx, dy, dz, qx, qy, qz);
    if (FL == 2 ? !((0.0f <= v) & (1.0f >= u + v)) : ((v < 0) | (u + v > 1))) return;
    const float tt = f * dot6<FL>(e2.x, e2.y, e2.z, qx, qy, qz);
    if (tt > 0.0001f && tt < dist) {
        dist = tt;
        triID = __float_as_uint(v0.w);
    }
}

// extend.cl:29-38 in the "shipped flags" flavour: t = (b - o) * v_rcp_f32(d) (the numerators arrive as b - o, the
// reciprocals in the ray's {d, rcp} pairs), then the same hardware min / max as box_fast
__device__ __forceinline__ bool box_shipped(float ax1, float ax2, float ay1, float ay2, float az1, float az2,
                                            float rx, float ry, float rz, float dist, float& tmin)
{
    const v2f tx = {ax1 * rx, ax2 * rx}, ty = {ay1 * ry, ay2 * ry}, tz = {az1 * rz, az2 * rz};
    return box_fast(tx, ty, tz, dist, tmin);
}

// In-place update of a loop-carried value inside a divergent branch: the write happens under the
// branch's exec mask into the SAME register, so hipcc has no second copy of the value to merge (it
// otherwise keeps a loop-carried and an in-body copy of the ray constants and moves one into the
// other on every trip).
__device__ __forceinline__ void set_in_place(float& dst, float v) { asm volatile("v_mov_b32 %0, %1" : "+v"(dst) : "v"(v)); }
__device__ __forceinline__ void set_in_place(uint32_t& dst, uint32_t v) { asm volatile("v_mov_b32 %0, %1" : "+v"(dst) : "v"(v)); }
__device__ __forceinline__ void set_in_place(int& dst, int v) { asm volatile("v_mov_b32 %0, %1" : "+v"(dst) : "v"(v)); }
__device__ __forceinline__ void set_in_place(v2f& dst, float lo, float hi)
{
    const v2f v = {lo, hi};
    asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "+v"(dst) : "v"(v));
}

struct Lane6 {
    v2f px, py, pz;         // {d, RN32(1/d)} per axis
    v2f po;                 // {origin y, dist}
    uint32_t triID;
    uint32_t cur;           // record reference: index | leaf bit + count code, REF_DONE = no ray
    int sp;
};

// One traversal step of one lane (extend.cl:44-80): an inner node (both children tested, ordered,
// descend / push / pop) or -- on a leaf trip -- a leaf (its triangles, pop).
// Stack entries 8..31 of this thread live in global memory (0.02 % of pushes on the test room).  The
// pointer is rebuilt from scratch where it is needed -- opaque to the compiler, which would otherwise
// keep it in two VGPRs (or a scratch slot) across the whole loop.
__device__ __forceinline__ uint32_t* ovf_ptr(const ExtendParams& p)
{
    uint32_t lane, wv;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    asm volatile("s_mov_b32 %0, %1" : "=s"(wv) : "s"(__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)));
    return p.ovf_stack + ((size_t)blockIdx.x * 256 + wv * 64 + lane) * (MAXS6 - PS6);
}

// the lane's number, computed where it is asked for (threadIdx.x, or a mbcnt the compiler can share, would stay live in a vector
// register from the kernel's entry on -- k_extend6 has none to spare)
__device__ __forceinline__ uint32_t lane_now()
{
    uint32_t lane;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
    return lane;
}

// the lane's rank among the lanes that are executing (asked for inside a divergent region: no mask to carry there)
__device__ __forceinline__ uint32_t lane_rank_in_exec()
{
    uint32_t r;
    asm volatile("v_mbcnt_lo_u32_b32 %0, exec_lo, 0\n\tv_mbcnt_hi_u32_b32 %0, exec_hi, %0" : "=v"(r));
    return r;
}

// The same for a ray that may have moved to another lane of its workgroup (k_extend6's drain merge): the rows belong to the lane whose
// LDS stack the ray uses.  stack_base = LDS address of that lane's entry -1; the stack array is aligned to its 1 KB rows.
__device__ __forceinline__ uint32_t* ovf_ptr(const ExtendParams& p, uint32_t stack_base)
{
    return p.ovf_stack + ((size_t)blockIdx.x * 256 + ((stack_base >> 2) & 255u)) * (MAXS6 - PS6);
}

}  // namespace uvrt
