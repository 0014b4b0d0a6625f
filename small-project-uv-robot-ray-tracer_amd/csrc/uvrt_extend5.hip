// uvrt_extend5.hip -- extend v5 (cl/extend.cl:6-99): the persistent-wave traversal of
// k_extend_persist (uvrt_kernels.hip) with two changes to the VALU work per loop trip, which is
// what bounds that kernel (profiles/r01_v4_*):
//
//  1. Slab distances by PACKED f32 arithmetic.  The reference's t = (b - o) / d needs the
//     correctly rounded binary32 quotient.  With a = RN32(b - o) and the per-ray constants
//     yh = RN32(1/d), yl = RN32(RN64(1/d) - yh):
//          t0 = RN(a * yl)
//          q0 = RN(a * yh + t0)          (fma)   -- a faithful rounding of a/d
//          r  = a - d * q0               (fma)   -- exact
//          q  = RN(q0 + r * yh)          (fma)   -- = RN32(a / d)   [Markstein's correction step]
//     Every step is a v_pk_*_f32, two slabs per instruction: 24 instructions per node pair instead
//     of the 48 (sub, cvt, mul_f64, cvt) of the f64-reciprocal form.  Proof and the exhaustive /
//     adversarial CPU check: DESIGN.md "Exact division, packed form", tests/test_recip_division.py.
//     Lanes outside the proof conditions (zero, > 1 or < 2^-60 direction component, tiny origin;
//     launches with tiny or huge scene bounds) take the reference's own IEEE division.
//  2. The x and z slabs subtract the LAMP's coordinates, which are launch-uniform
//     (generate.cl:16): k_prepare_launch writes a per-launch copy of the node-pair records with
//     a = RN32(b - o) already applied to x and z (the same single f32 subtraction), laid out so
//     that the (min, max) bounds of one axis of one child are a register pair.
//  3. Leaf visits are made on every LEAFP-th trip only (lanes standing at a leaf wait): the
//     triangle test runs for ~8 % of the lanes but costs a third of a trip's instructions.
//
// Visit order, every AABB / triangle test and every comparison are the reference's.
#include "uvrt_device.h"

namespace uvrt {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int MAX_STACK5 = 32;   // extend.cl:43
constexpr int PSTACK5 = 8;       // LDS stack entries per lane

// Per-ray state.  (direction component, its reciprocal) and (origin y, closest distance) are
// register PAIRS: a packed-f32 instruction broadcasts either half of a pair with op_sel.
struct Ray5 {
    float ox, oz;           // launch-uniform (SGPRs); used by the triangle test only
    v2f px, py, pz;         // {d, RN32(1/d)} per axis
    v2f po;                 // {origin y, dist}
    uint32_t triID;
};

// Two correctly rounded quotients {a.x / d, a.y / d}, dy = {d, RN32(1/d)} (see the file header):
//   q0 = a * y ; r = a - d * q0 (exact) ; q = q0 + r * y.
// Written as inline asm so that each step is exactly one packed instruction with the broadcast
// done by op_sel (hipcc builds the splat operands with extra v_mov / v_xor otherwise).
__device__ __forceinline__ v2f div2(v2f a, v2f dy)
{
    v2f q0, r, q;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(q0) : "v"(a), "v"(dy));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]"
        : "=v"(r) : "v"(dy), "v"(q0), "v"(a));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(q) : "v"(r), "v"(dy), "v"(q0));
    return q;
}

// {a.x - o.x, a.y - o.x}: the y slab's numerators (extend.cl:33), o = {origin y, .}
__device__ __forceinline__ v2f sub_lo2(v2f a, v2f o)
{
    v2f d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(o));
    return d;
}

// extend.cl:29-38 from the three (t at min, t at max) pairs.  No operand is NaN on this path, so
// the hardware min/max equal OpenCL's y<x?y:x / x<y?y:x; asm, because fminf/fmaxf on values that
// come out of inline asm get a canonicalising v_max_f32 x, x each.
__device__ __forceinline__ float box_fast(v2f tx, v2f ty, v2f tz, float dist)
{
    float nx, fx, ny, fy, nz, fz, tmin, tmax;
    asm("v_min_f32 %0, %1, %2" : "=v"(nx) : "v"(tx.x), "v"(tx.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fx) : "v"(tx.x), "v"(tx.y));
    asm("v_min_f32 %0, %1, %2" : "=v"(ny) : "v"(ty.x), "v"(ty.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fy) : "v"(ty.x), "v"(ty.y));
    asm("v_min_f32 %0, %1, %2" : "=v"(nz) : "v"(tz.x), "v"(tz.y));
    asm("v_max_f32 %0, %1, %2" : "=v"(fz) : "v"(tz.x), "v"(tz.y));
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tmin) : "v"(nx), "v"(ny), "v"(nz));   // max(max(nx, ny), nz)
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tmax) : "v"(fx), "v"(fy), "v"(fz));   // min(min(fx, fy), fz)
    const bool hit = (tmax >= tmin) & (tmin < dist) & (tmax > 0);
    return hit ? tmin : 1e30f;
}

// the reference's own form: IEEE divisions, OpenCL min/max as selects (NaN operands: 0/0)
__device__ __forceinline__ float box_exact(float ax1, float ax2, float ay1, float ay2, float az1, float az2,
                                           const Ray5& r)
{
    const float tx1 = ax1 / r.px.x, tx2 = ax2 / r.px.x;
    float tmin = tx2 < tx1 ? tx2 : tx1, tmax = tx1 < tx2 ? tx2 : tx1;
    const float ty1 = ay1 / r.py.x, ty2 = ay2 / r.py.x;
    const float mny = ty2 < ty1 ? ty2 : ty1, mxy = ty1 < ty2 ? ty2 : ty1;
    tmin = tmin < mny ? mny : tmin;
    tmax = mxy < tmax ? mxy : tmax;
    const float tz1 = az1 / r.pz.x, tz2 = az2 / r.pz.x;
    const float mnz = tz2 < tz1 ? tz2 : tz1, mxz = tz1 < tz2 ? tz2 : tz1;
    tmin = tmin < mnz ? mnz : tmin;
    tmax = mxz < tmax ? mxz : tmax;
    if (tmax >= tmin && tmin < r.po.y && tmax > 0) return tmin;
    return 1e30f;
}

// extend.cl:6-27 on a leaf record (v0, e1 = v1 - v0, e2 = v2 - v0, id in v0.w)
__device__ __forceinline__ void tri5(Ray5& r, const float4 v0, const float4 e1, const float4 e2)
{
    const float dx = r.px.x, dy = r.py.x, dz = r.pz.x;
    const float hx = dy * e2.z - dz * e2.y;
    const float hy = dz * e2.x - dx * e2.z;
    const float hz = dx * e2.y - dy * e2.x;
    const float a = e1.x * hx + e1.y * hy + e1.z * hz;
    if (fabsf(a) < 0.00001f) return;
    const float f = 1.0f / a;
    const float sx = r.ox - v0.x, sy = r.po.x - v0.y, sz = r.oz - v0.z;
    const float u = f * (sx * hx + sy * hy + sz * hz);
    if ((u < 0) | (u > 1)) return;
    const float qx = sy * e1.z - sz * e1.y;
    const float qy = sz * e1.x - sx * e1.z;
    const float qz = sx * e1.y - sy * e1.x;
    const float v = f * (dx * qx + dy * qy + dz * qz);
    if ((v < 0) | (u + v > 1)) return;
    const float tt = f * (e2.x * qx + e2.y * qy + e2.z * qz);
    if (tt > 0.0001f && tt < r.po.y) {
        r.po.y = tt;
        r.triID = __float_as_uint(v0.w);
    }
}

// One traversal step of one lane (extend.cl:44-80): an inner node (both children tested, ordered,
// descend / push / pop) or -- when `leaf_trip` -- a leaf (its triangles, pop).  Inner and leaf
// lanes share ONE set of four 16-byte loads (per-lane address select), as in k_extend_persist.
// Top-of-tree cache: the first `top_pairs` per-launch records (breadth-first numbering = the upper
// tree levels, ~40 % of all node visits) are copied into LDS by every workgroup; part q of record
// rec lies in 16-byte slot q ^ ((rec >> 2) & 3) of its 64-byte block, which spreads the lanes of
// a ds_read_b128 over sixteen 4-bank windows instead of four.
constexpr uint32_t TOP5_MAX = 127;        // 7 complete levels, 8 KB of LDS

template <bool EXACT, bool TOP>
__device__ __forceinline__ void step5(Ray5& r, uint32_t& cur, int& sp, uint32_t* ovf, const ExtendParams& p,
                                      uint32_t (*s_stack)[256], const float4* s_top, uint32_t top_pairs,
                                      bool leaf_trip)
{
    const int tid = threadIdx.x;
    const bool is_inner = cur < REF_LEAF_BIT;
    const bool is_leaf = !is_inner && cur != REF_DONE && leaf_trip;
    const uint32_t first = cur & REF_FIRST_MASK;
    const char* base = (const char*)p.lpairs;
    const int64_t off = is_inner ? (int64_t)cur * 64
                                 : ((const char*)p.scene.ltris - base) + (int64_t)first * (int64_t)sizeof(LeafTri);
    v4f w0, w1, w2, w3;   // written by the loads below, read only by the lanes that executed them
    uint32_t spec_top = REF_DONE;
    const bool in_top = TOP && cur < top_pairs;
    if (in_top) {
        // LDS byte address of slot 0 ^ s; the other three parts are at ^16, ^32, ^48
        const uint32_t a0 = (uint32_t)(uintptr_t)s_top + (cur << 6) + ((cur & 12u) << 2);
        const uint32_t a1 = a0 ^ 16u, a2 = a0 ^ 32u, a3 = a0 ^ 48u;
        if (sp > 0 && sp <= PSTACK5) spec_top = s_stack[sp - 1][tid];
        asm volatile("ds_read_b128 %0, %4\n\t"
                     "ds_read_b128 %1, %5\n\t"
                     "ds_read_b128 %2, %6\n\t"
                     "ds_read_b128 %3, %7\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3)
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3)
                     : "memory");
    } else if (is_inner | is_leaf) {
        if (sp > 0 && sp <= PSTACK5) spec_top = s_stack[sp - 1][tid];
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
        float dist1, dist2;
        if (EXACT) {
            dist1 = box_exact(w0.x, w0.y, w2.x - r.po.x, w2.y - r.po.x, w0.z, w0.w, r);
            dist2 = box_exact(w1.x, w1.y, w2.z - r.po.x, w2.w - r.po.x, w1.z, w1.w, r);
        } else {
            const v2f x0 = __builtin_shufflevector(w0, w0, 0, 1), z0 = __builtin_shufflevector(w0, w0, 2, 3);
            const v2f x1 = __builtin_shufflevector(w1, w1, 0, 1), z1 = __builtin_shufflevector(w1, w1, 2, 3);
            const v2f y0 = sub_lo2(__builtin_shufflevector(w2, w2, 0, 1), r.po);
            const v2f y1 = sub_lo2(__builtin_shufflevector(w2, w2, 2, 3), r.po);
            dist1 = box_fast(div2(x0, r.px), div2(y0, r.py), div2(z0, r.pz), r.po.y);
            dist2 = box_fast(div2(x1, r.px), div2(y1, r.py), div2(z1, r.pz), r.po.y);
        }
        uint32_t ref1 = __float_as_uint(w3.x), ref2 = __float_as_uint(w3.y);
        if (dist1 > dist2) {                               // extend.cl:61-65
            const float td = dist1; dist1 = dist2; dist2 = td;
            const uint32_t tr = ref1; ref1 = ref2; ref2 = tr;
        }
        if (dist1 == 1e30f) pop = true;                    // :66-69
        else {                                             // :70-76
            cur = ref1;
            if (dist2 != 1e30f) {
                if (sp < PSTACK5) s_stack[sp][tid] = ref2;
                else if (sp < MAX_STACK5) ovf[sp - PSTACK5] = ref2;
                else *p.error_flag = 1u;
                if (sp < MAX_STACK5) ++sp;
            }
        }
    } else if (is_leaf) {                                  // extend.cl:48-55
        uint32_t count = (cur >> REF_COUNT_SHIFT) & 15u;
        if (count == 15u) count = p.scene.leaf_count[first];
        tri5(r, make_float4(w0.x, w0.y, w0.z, w0.w), make_float4(w1.x, w1.y, w1.z, w1.w),
             make_float4(w2.x, w2.y, w2.z, w2.w));
        for (uint32_t i = 1; i < count; ++i) {
            const LeafTri* lt = p.scene.ltris + first + i;
            tri5(r, lt->v0_id, lt->e1, lt->e2);
        }
        pop = true;
    }
    if (pop) {
        if (sp == 0) cur = REF_DONE;
        else {
            --sp;
            if (sp < PSTACK5) cur = spec_top;
            else cur = ovf[sp - PSTACK5];
        }
    }
}

template <int REFILL_MIN, int LEAFP, bool RECORD, bool TOP>
__global__ __launch_bounds__(256, 8) void k_extend5(ExtendParams p)
{
    __shared__ uint32_t s_stack[PSTACK5][256];                      // 8 KB
    __shared__ float4 s_top[TOP ? (TOP5_MAX + 1) * 4 : 4];          // 8 KB
    const uint32_t top_pairs = TOP ? (p.top_pairs < TOP5_MAX ? p.top_pairs : TOP5_MAX) : 0u;
    if (TOP) {
        const float4* src = (const float4*)p.lpairs;
        for (uint32_t i = threadIdx.x; i < top_pairs * 4u; i += 256u) {
            const uint32_t rec = i >> 2;
            s_top[rec * 4u + ((i & 3u) ^ ((rec >> 2) & 3u))] = src[i];
        }
        __syncthreads();
    }
    uint32_t* const ovf = p.ovf_stack + ((size_t)blockIdx.x * 256 + threadIdx.x) * (MAX_STACK5 - PSTACK5);
    Ray5 r;
    r.ox = p.ox; r.oz = p.oz;
    r.px = r.py = r.pz = (v2f){1.f, 1.f};
    r.po = (v2f){0.f, 1e30f};
    r.triID = 0;
    uint32_t cur = REF_DONE;   // this lane holds no ray
    uint32_t slot = 0;
    int sp = 0;
    bool live = false;         // holds a ray whose result has not been deposited yet
    bool special = false;      // this lane's ray needs the EXACT path
    int32_t* const my_counts = p.counts + (int64_t)(blockIdx.x % (unsigned)p.count_replicas) * p.count_stride;

    // wave w traces the 64-ray batches w, w + W, w + 2W, ... (see k_extend_persist)
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const uint32_t W = gridDim.x * 4u;
    uint32_t cursor = 0;
    const uint32_t chunk_end = p.chunk;
    const uint32_t n32 = (uint32_t)p.n;
    uint32_t trip = 0;

    for (;;) {
        const bool idle = cur == REF_DONE;
        const unsigned long long idle_mask = __ballot(idle);
        const int nidle = __popcll(idle_mask);
        if (cursor < chunk_end && nidle >= REFILL_MIN) {
            if (idle) {
                if (live) {                                 // deposit what this lane finished
                    live = false;
                    if (RECORD && p.hits) {
                        const uint32_t li = p.order ? p.order[slot] : slot;
                        p.hits[li] = make_uint2(__float_as_uint(r.po.y), r.triID);
                    }
                    if (r.po.y != 1e30f) atomicAdd(&my_counts[r.triID], 1);   // extend.cl:94-98
                }
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                      __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const uint32_t v = cursor + rank;
                const uint32_t my = ((v >> 6) * W + wave) * 64u + (v & 63u);
                if (v < chunk_end && my < n32) {
                    const float4 rec = p.rays[my];
                    // RN32(RN64(1/d)) = RN32(1/d): 1/d is never within 2^-49 of a binary32 midpoint
                    r.px = (v2f){rec.x, (float)p.recip[my]};
                    r.py = (v2f){rec.y, (float)p.recip[p.recip_stride + my]};
                    r.pz = (v2f){rec.z, (float)p.recip[2 * p.recip_stride + my]};
                    r.po = (v2f){rec.w, 1e30f};            // generate.cl:34-35
                    r.triID = 0;
                    if (RECORD) slot = my;
                    sp = 0;
                    cur = p.scene.root_ref;
                    live = true;
                    const float ay = fabsf(rec.w), adx = fabsf(rec.x), ady = fabsf(rec.y), adz = fabsf(rec.z);
                    const float dmin = 8.6736174e-19f;     // 2^-60 (also catches zero components)
                    special = !(adx >= dmin) || !(ady >= dmin) || !(adz >= dmin) ||
                              !(adx <= 1.0f) || !(ady <= 1.0f) || !(adz <= 1.0f) ||
                              (ay != 0.0f && ay < 7.888609e-31f) || !(ay <= 1e18f) || p.force_exact != 0;
                }
            }
            cursor += (uint32_t)nidle;
        }
        const bool active = cur != REF_DONE;
        if (!__any(active)) {
            if (cursor >= chunk_end) break;
            continue;
        }
        bool leaf_trip = true;
        if (LEAFP > 1) {
            leaf_trip = (trip % (uint32_t)LEAFP) == 0u || !__any(cur < REF_LEAF_BIT);
            ++trip;
        }
        if (__any(active & special)) step5<true, TOP>(r, cur, sp, ovf, p, s_stack, s_top, top_pairs, leaf_trip);
        else step5<false, TOP>(r, cur, sp, ovf, p, s_stack, s_top, top_pairs, leaf_trip);
    }
    if (live) {
        if (RECORD && p.hits) {
            const uint32_t li = p.order ? p.order[slot] : slot;
            p.hits[li] = make_uint2(__float_as_uint(r.po.y), r.triID);
        }
        if (r.po.y != 1e30f) atomicAdd(&my_counts[r.triID], 1);       // extend.cl:94-98
    }
}

// Per-launch node-pair records: the lamp's x and z subtracted from the x / z bounds (the same
// single f32 subtraction IntersectAABB performs, extend.cl:31,35), one axis of one child per
// register pair.  w0 = child0 {minx, maxx, minz, maxz}, w1 = child1 likewise,
// w2 = {c0 miny, c0 maxy, c1 miny, c1 maxy} (raw), w3 = {ref0, ref1, 0, 0}.
__global__ __launch_bounds__(256) void k_prepare_launch(const PairRec* __restrict__ pairs,
                                                        float4* __restrict__ lpairs, float ox, float oz,
                                                        int32_t npairs)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    const PairRec pr = pairs[i];
    lpairs[i * 4 + 0] = make_float4(pr.c0min_ref0.x - ox, pr.c0max_ref1.x - ox, pr.c0min_ref0.z - oz, pr.c0max_ref1.z - oz);
    lpairs[i * 4 + 1] = make_float4(pr.c1min.x - ox, pr.c1max.x - ox, pr.c1min.z - oz, pr.c1max.z - oz);
    lpairs[i * 4 + 2] = make_float4(pr.c0min_ref0.y, pr.c0max_ref1.y, pr.c1min.y, pr.c1max.y);
    lpairs[i * 4 + 3] = make_float4(pr.c0min_ref0.w, pr.c0max_ref1.w, 0.f, 0.f);
}

// leaf_code: 0..3 -> leaf visits every 1/2/3/4 trips, +4 -> with the top-of-tree LDS cache
bool launch_extend5(const ExtendParams& p0, int leaf_code, int grid_per_cu, hipStream_t s)
{
    if (p0.n <= 0) return true;
    ExtendParams p = p0;
    const unsigned cus = p.num_cus > 0 ? (unsigned)p.num_cus : 256u;
    unsigned grid = cus * (unsigned)grid_per_cu;
    const unsigned need = (unsigned)((p.n + 255) / 256);
    if (need < grid) grid = need;
    const uint64_t waves = (uint64_t)grid * 4;
    p.chunk = (uint32_t)((((uint64_t)p.n + waves - 1) / waves + 63) / 64 * 64);
    if ((uint64_t)grid * 256 * (MAX_STACK5 - PSTACK5) > p.ovf_capacity) return false;
    if (p.npairs > 0)
        hipLaunchKernelGGL(k_prepare_launch, dim3((unsigned)((p.npairs + 255) / 256)), dim3(256), 0, s,
                           p.scene.pairs, (float4*)p.lpairs, p.ox, p.oz, p.npairs);
#define UVRT_L5(LP, TOP)                                                                                \
    do {                                                                                                \
        if (p.hits) hipLaunchKernelGGL((k_extend5<16, LP, true, TOP>), dim3(grid), dim3(256), 0, s, p);    \
        else hipLaunchKernelGGL((k_extend5<16, LP, false, TOP>), dim3(grid), dim3(256), 0, s, p);         \
    } while (0)
    switch (leaf_code) {
        case 1: UVRT_L5(2, false); break;
        case 2: UVRT_L5(3, false); break;
        case 3: UVRT_L5(4, false); break;
        case 4: UVRT_L5(1, true); break;
        case 5: UVRT_L5(2, true); break;
        case 6: UVRT_L5(3, true); break;
        case 7: UVRT_L5(4, true); break;
        default: UVRT_L5(1, false); break;
    }
#undef UVRT_L5
    return true;
}

}  // namespace uvrt
