// uvrt_extend4.hip -- extend over the opt-in 4-WIDE collapse of the reference's BVH.
//
// uvrt_set_wide_bvh(ctx, 1) collapses the flat BVH2 the caller hands over (bvh.cpp output) by one level:
// a 4-wide node holds the boxes of its BVH2 node's grandchildren (or of a child that is a leaf).  Every box
// and every triangle test is the SAME arithmetic as in uvrt_extend6.hip (exact slab distances, the
// reference's Moeller-Trumbore), only fewer boxes are tested (the two intermediate child boxes of a node are
// skipped; by monotonicity of correctly rounded (b - o) / d in b a ray that enters a grandchild's box also
// enters the child's) and the visit order differs: the nearest hit child first, the others pushed in slot
// order.  Closest hit is order-independent except where two accepted hits have exactly equal t
// (extend.cl:25 is a strict <: first found wins) or where a box is culled by a hit found earlier at an
// almost equal distance, so `dist` and `triID` equal the reference's on all but such rays
// (tests/test_gpu_wide_bvh.py counts them: none on the test room in 25 M rays).  It is therefore OPT-IN: the
// default kernel keeps the reference's visit order and is bit-exact unconditionally.
//
// Per trip a lane fetches ONE 128-byte record (= one L2 line: seven dwordx4 loads, eight lookups in one L1
// line instead of two BVH2 records in two lines) and makes at most one descent and three pushes; a ray
// needs about half the trips of the BVH2 walk.
#include "uvrt_traverse.h"

namespace uvrt {

constexpr uint32_t TOP4_MAX = 64;         // nodes cached in LDS
constexpr uint32_t TOP4_STRIDE = 144;     // bytes per cached node (128 + 16 padding): 9 KB
constexpr uint32_t KEY_MISS = 0xFFFFFFFFu;

template <int FL>
__device__ __forceinline__ void step4(Lane6& L, const ExtendParams& p, uint32_t stack_base, const float4* s_top,
                                      uint32_t top_units, bool leaf_trip, bool exact, unsigned long long m_act)
{
    const uint32_t cur = L.cur;
    const bool is_inner = cur < REF_LEAF_BIT;
    const bool is_leaf = (cur >= REF_LEAF_BIT) & (cur != REF_DONE) & leaf_trip;
    const uint32_t idx = cur & REF_FIRST_MASK;          // 64-byte unit index of the record
    v4f w0, w1, w2, w3, w4, w5, w6;
    uint32_t spec_top = REF_DONE;
    const uint32_t sa = stack_base + ((uint32_t)L.sp << 10);
    {
        const unsigned long long m_in = __builtin_amdgcn_ballot_w64(cur < REF_LEAF_BIT);
        const unsigned long long m_top = __builtin_amdgcn_ballot_w64(cur < top_units);
        const unsigned long long m_sp = __builtin_amdgcn_ballot_w64(L.sp > 0);
        const unsigned long long m_go = m_in | (leaf_trip ? (m_act & ~m_in) : 0ull);
        const unsigned long long m_glob = m_go & ~m_top;            // first 48 bytes: inner and leaf lanes
        const unsigned long long m_glob_in = m_in & ~m_top;         // the rest of a node record
        const unsigned long long m_stk = m_go & m_sp;
        const uint32_t a0 = (uint32_t)(uintptr_t)s_top + (cur >> 1) * TOP4_STRIDE;
        const uint32_t roff = cur << 6;
        unsigned long long save;
        asm volatile("s_mov_b64 %[save], exec\n\t"
                     "s_mov_b64 exec, %[mstk]\n\t"
                     "ds_read_b32 %[st], %[sa]\n\t"
                     "s_mov_b64 exec, %[mtop]\n\t"
                     "ds_read_b128 %[w0], %[a0]\n\t"
                     "ds_read_b128 %[w1], %[a0] offset:16\n\t"
                     "ds_read_b128 %[w2], %[a0] offset:32\n\t"
                     "ds_read_b128 %[w3], %[a0] offset:48\n\t"
                     "ds_read_b128 %[w4], %[a0] offset:64\n\t"
                     "ds_read_b128 %[w5], %[a0] offset:80\n\t"
                     "ds_read_b128 %[w6], %[a0] offset:96\n\t"
                     "s_mov_b64 exec, %[mglob]\n\t"
                     "global_load_dwordx4 %[w0], %[ro], %[rb]\n\t"
                     "global_load_dwordx4 %[w1], %[ro], %[rb] offset:16\n\t"
                     "global_load_dwordx4 %[w2], %[ro], %[rb] offset:32\n\t"
                     "s_mov_b64 exec, %[mgin]\n\t"
                     "global_load_dwordx4 %[w3], %[ro], %[rb] offset:48\n\t"
                     "global_load_dwordx4 %[w4], %[ro], %[rb] offset:64\n\t"
                     "global_load_dwordx4 %[w5], %[ro], %[rb] offset:80\n\t"
                     "global_load_dwordx4 %[w6], %[ro], %[rb] offset:96\n\t"
                     "s_mov_b64 exec, %[save]\n\t"
                     "s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [w4] "=&v"(w4), [w5] "=&v"(w5),
                       [w6] "=&v"(w6), [st] "+v"(spec_top), [save] "=&s"(save)
                     : [a0] "v"(a0), [sa] "v"(sa), [ro] "v"(roff), [rb] "s"(p.recs4), [mtop] "s"(m_top), [mglob] "s"(m_glob),
                       [mgin] "s"(m_glob_in), [mstk] "s"(m_stk)
                     : "memory");
    }
    bool need_pop = is_leaf;
    if (is_leaf) {                                         // extend.cl:48-55
        uint32_t count = (cur >> REF_COUNT_SHIFT) & 15u;
        const uint32_t first = idx - 2u * (uint32_t)p.nquads;
        if (count == 15u) count = p.scene.leaf_count[first];
        float dist = L.po.y;
        tri6<FL>(p.ox, L.po.x, p.oz, L.px.x, L.py.x, L.pz.x, dist, L.triID, make_float4(w0.x, w0.y, w0.z, w0.w),
                  make_float4(w1.x, w1.y, w1.z, w1.w), make_float4(w2.x, w2.y, w2.z, w2.w), exact);
        for (uint32_t i = 1; i < count; ++i) {
            const float4* lt = (const float4*)p.recs4 + ((size_t)idx + i) * 4;
            tri6<FL>(p.ox, L.po.x, p.oz, L.px.x, L.py.x, L.pz.x, dist, L.triID, lt[0], lt[1], lt[2], exact);
        }
        L.po.y = dist;
    }
    if (is_inner) {
        float d[4];
        bool h[4];
        const v4f xz[4] = {w0, w1, w2, w3};
        const v2f yy[4] = {__builtin_shufflevector(w4, w4, 0, 1), __builtin_shufflevector(w4, w4, 2, 3),
                           __builtin_shufflevector(w5, w5, 0, 1), __builtin_shufflevector(w5, w5, 2, 3)};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (FL == 2) {       // "shipped flags": t = (b - o) * v_rcp_f32(d) (uvrt_traverse.h slabs6s)
                v2f x = __builtin_shufflevector(xz[k], xz[k], 0, 1), z = __builtin_shufflevector(xz[k], xz[k], 2, 3), y = yy[k];
                slabs6s(x, y, z, L.px, L.py, L.pz, L.po);
                h[k] = box_fast(x, y, z, L.po.y, d[k]);
            } else if (exact) {
                h[k] = box_exact(xz[k].x, xz[k].y, yy[k].x - L.po.x, yy[k].y - L.po.x, xz[k].z, xz[k].w, L.px.x, L.py.x,
                                 L.pz.x, L.po.y, d[k]);
            } else {
                v2f x = __builtin_shufflevector(xz[k], xz[k], 0, 1), z = __builtin_shufflevector(xz[k], xz[k], 2, 3), y = yy[k];
                slabs6(x, y, z, L.px, L.py, L.pz, L.po);
                h[k] = box_fast(x, y, z, L.po.y, d[k]);
            }
        }
        // nearest hit child first: the entry distances as ordered integers (a negative one -- the origin is
        // inside the box -- counts as 0) with the slot number in the two lowest bits
        uint32_t key[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            key[k] = h[k] ? ((__float_as_uint(fmaxf(d[k], 0.0f)) & ~3u) | (uint32_t)k) : KEY_MISS;
        const uint32_t kmin = min(min(key[0], key[1]), min(key[2], key[3]));
        const uint32_t near = kmin & 3u;
        const uint32_t r[4] = {__float_as_uint(w6.x), __float_as_uint(w6.y), __float_as_uint(w6.z), __float_as_uint(w6.w)};
        need_pop = kmin == KEY_MISS;
        if (!need_pop) {
#pragma unroll
            for (int k = 3; k >= 0; --k) {                 // the others wait on the stack (extend.cl:76)
                if (h[k] & (near != (uint32_t)k)) {
                    const uint32_t sak = stack_base + ((uint32_t)L.sp << 10);
                    if (L.sp < PS6) asm volatile("ds_write_b32 %0, %1 offset:1024" : : "v"(sak), "v"(r[k]) : "memory");
                    else if (L.sp < MAXS6) ovf_ptr(p)[L.sp - PS6] = r[k];
                    else *p.error_flag = 1u;
                    L.sp = L.sp < MAXS6 ? L.sp + 1 : L.sp;
                }
            }
            L.cur = near == 0u ? r[0] : near == 1u ? r[1] : near == 2u ? r[2] : r[3];
        }
    }
    if (need_pop) {
        uint32_t popped = spec_top;
        if (L.sp > PS6) popped = ovf_ptr(p)[L.sp - 1 - PS6];
        L.cur = popped;
        L.sp = (int)__builtin_elementwise_sub_sat((uint32_t)L.sp, 1u);
    }
}

template <bool RECORD, int FL>
__global__ __launch_bounds__(256, 6) void k_extend4(ExtendParams p)
{
    __shared__ uint32_t s_stack[PS6][256];                              // 8 KB
    __shared__ float4 s_top[(TOP4_MAX + 1) * 9];                        // 9 KB
    const uint32_t top_quads = p.top_quads < TOP4_MAX ? p.top_quads : TOP4_MAX;
    {
        const float4* src = (const float4*)p.recs4;
        for (uint32_t i = threadIdx.x; i < top_quads * 8u; i += 256u) s_top[(i >> 3) * 9u + (i & 7u)] = src[i];
        __syncthreads();
    }
    const uint32_t stack_base = (uint32_t)(uintptr_t)&s_stack[0][threadIdx.x] - 1024u;
    Lane6 L;
    L.px = L.py = L.pz = (v2f){1.f, 1.f};
    L.po = (v2f){0.f, 1e30f};
    L.triID = 0;
    L.cur = REF_DONE;
    L.sp = 0;
    uint32_t slot = 0;
    bool live = false;
    unsigned long long special_mask = 0;
    int32_t* const my_counts = p.counts + (int64_t)(blockIdx.x % (unsigned)p.count_replicas) * p.count_stride;
    uint32_t plane_off = 0;
    const float plane_inv = p.plane_inv;
    const uint32_t wave = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t W = gridDim.x * 4u;
    uint32_t cursor = 0;
    const uint32_t chunk_end = p.chunk;
    const uint32_t n32 = (uint32_t)p.n;
    uint32_t trip = 0;

    for (;;) {
        const unsigned long long idle_mask = __builtin_amdgcn_ballot_w64(L.cur == REF_DONE);
        const int nidle = __popcll(idle_mask);
        if (cursor < chunk_end && nidle >= p.refill_min) {
            bool spec = false;
            if (L.cur == REF_DONE) {
                if (RECORD && live && p.hits) {
                    const uint32_t li = p.order ? p.order[slot] : slot;
                    p.hits[li] = make_uint2(__float_as_uint(L.po.y), L.triID);
                }
                if (L.po.y != 1e30f) atomicAdd(&my_counts[plane_off + L.triID], 1);
                live = false;
                L.po.y = 1e30f;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                      __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                const uint32_t v = cursor + rank;
                const uint32_t gb = (v >> 6) * W + wave;
                const uint32_t my = gb * 64u + (v & 63u);
                uint32_t pl = (uint32_t)((float)gb * plane_inv);
                int32_t within = (int32_t)(gb - pl * p.plane_batches);
                if (within < 0) { --pl; within += (int32_t)p.plane_batches; }
                else if ((uint32_t)within >= p.plane_batches) { ++pl; within -= (int32_t)p.plane_batches; }
                if (v < chunk_end && my < n32 && (uint32_t)within * 64u + (v & 63u) < p.plane_n) {
                    set_in_place(plane_off, pl * p.plane_stride);
                    const float4 rec = p.rays[my];
                    set_in_place(L.px, rec.x, FL == 2 ? rcp_raw(rec.x) : rcp_exact(rec.x));
                    set_in_place(L.py, rec.y, FL == 2 ? rcp_raw(rec.y) : rcp_exact(rec.y));
                    set_in_place(L.pz, rec.z, FL == 2 ? rcp_raw(rec.z) : rcp_exact(rec.z));
                    set_in_place(L.po, rec.w, 1e30f);
                    set_in_place(L.triID, 0u);
                    if (RECORD) { slot = my; live = true; }
                    set_in_place(L.sp, 0);
                    set_in_place(L.cur, p.root_ref4);
                    const float ay = fabsf(rec.w), adx = fabsf(rec.x), ady = fabsf(rec.y), adz = fabsf(rec.z);
                    const float dmin = 8.6736174e-19f;
                    spec = FL != 2 && (!(adx >= dmin) || !(ady >= dmin) || !(adz >= dmin) ||
                           !(adx <= 1.0f) || !(ady <= 1.0f) || !(adz <= 1.0f) ||
                           (ay != 0.0f && ay < 7.888609e-31f) || !(ay <= 1e9f) || p.force_exact != 0);
                }
            }
            cursor += (uint32_t)nidle;
            special_mask = (special_mask & ~idle_mask) | __builtin_amdgcn_ballot_w64(spec);
        }
        const unsigned long long act = __builtin_amdgcn_ballot_w64(L.cur != REF_DONE);
        if (act == 0) {
            if (cursor >= chunk_end) break;
            continue;
        }
        const bool leaf_trip = (trip & 1u) == 0u || __builtin_amdgcn_ballot_w64(L.cur < REF_LEAF_BIT) == 0;
        ++trip;
        step4<FL>(L, p, stack_base, s_top, 2u * top_quads, leaf_trip, (special_mask & act) != 0, act);
    }
    if (RECORD && live && p.hits) {
        const uint32_t li = p.order ? p.order[slot] : slot;
        p.hits[li] = make_uint2(__float_as_uint(L.po.y), L.triID);
    }
    if (L.po.y != 1e30f) atomicAdd(&my_counts[plane_off + L.triID], 1);
}

// per-launch node records: the lamp's x and z subtracted from the x / z bounds (extend.cl:31,35), leaf
// references re-based to record units (+ 2 * nquads)
__global__ __launch_bounds__(256) void k_prepare_launch4(const QuadRec* __restrict__ quads, QuadRec* __restrict__ recs,
                                                         float ox, float oz, int32_t nquads)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nquads) return;
    QuadRec q = quads[i];
    for (int k = 0; k < 4; ++k) {
        q.xz[k] = make_float4(q.xz[k].x - ox, q.xz[k].y - ox, q.xz[k].z - oz, q.xz[k].w - oz);
        if (q.ref[k] >= REF_LEAF_BIT && q.ref[k] != REF_DONE) q.ref[k] += 2u * (uint32_t)nquads;
    }
    recs[i] = q;
}

void launch_prepare_launch4(const QuadRec* quads, void* recs4, float ox, float oz, int32_t nquads, hipStream_t s)
{
    if (nquads <= 0) return;
    hipLaunchKernelGGL(k_prepare_launch4, dim3((unsigned)((nquads + 255) / 256)), dim3(256), 0, s, quads, (QuadRec*)recs4,
                       ox, oz, nquads);
}

bool launch_extend4(const ExtendParams& p0, int grid_per_cu, hipStream_t s)
{
    if (p0.n <= 0) return true;
    ExtendParams p = p0;
    const unsigned cus = p.num_cus > 0 ? (unsigned)p.num_cus : 256u;
    unsigned grid = cus * (unsigned)grid_per_cu;
    if (p.plane_batches == 0) {
        p.plane_batches = (uint32_t)((p.n + 63) / 64);
        p.plane_n = (uint32_t)p.n;
        p.plane_stride = 0;
    }
    p.plane_inv = 1.0f / (float)p.plane_batches;
    const unsigned need = (unsigned)((p.n + 255) / 256);
    if (need < grid) grid = need;
    const uint64_t waves = (uint64_t)grid * 4;
    p.chunk = (uint32_t)((((uint64_t)p.n + waves - 1) / waves + 63) / 64 * 64);
    if ((uint64_t)grid * 256 * (MAXS6 - PS6) > p.ovf_capacity) return false;
    p.root_ref4 = (p.scene.root_ref >= REF_LEAF_BIT && p.scene.root_ref != REF_DONE)
                      ? p.scene.root_ref + 2u * (uint32_t)p.nquads : p.scene.root_ref;
    if (p.flavour == 2) {
        if (p.hits) hipLaunchKernelGGL((k_extend4<true, 2>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_extend4<false, 2>), dim3(grid), dim3(256), 0, s, p);
    } else if (p.flavour) {
        if (p.hits) hipLaunchKernelGGL((k_extend4<true, 1>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_extend4<false, 1>), dim3(grid), dim3(256), 0, s, p);
    } else {
        if (p.hits) hipLaunchKernelGGL((k_extend4<true, 0>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((k_extend4<false, 0>), dim3(grid), dim3(256), 0, s, p);
    }
    return true;
}

}  // namespace uvrt
