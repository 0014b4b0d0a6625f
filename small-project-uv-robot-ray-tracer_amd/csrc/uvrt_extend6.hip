// uvrt_extend6.hip -- extend (cl/extend.cl:6-99), the traversal kernel ("v6": the sixth generation;
// DESIGN.md section 4 has what its predecessors measured).
//
// Persistent wavefronts over statically owned 64-ray batches, in-wave refill, one traversal step per
// lane and loop trip, the reference's visit order / tests / comparisons.  The instruction stream of a
// trip is what the design is about (the launch is latency-bound with VALU issue, scalar
// issue and the L1's lookup rate all at 45-55 % of their calibrated peaks):
//
//  * slab distances t = (b - o) / d as PACKED f32, two quotients per instruction:
//        q0 = a * y;  r = fma(-d, q0, a);  q = fma(r, y, q0)      with y = RN32(1/d)
//    equals RN32(a / d) for all normal operands -- proven by exhaustion over all 2^46 significand
//    pairs on the GPU (tests/tools/div3_exhaustive.hip, profiles/r01/r01_div3_exhaustive.log);
//  * f = 1 / a of the triangle test (extend.cl:17) as v_rcp_f32 + one Newton step, which is the
//    correctly rounded reciprocal for every binary32 in [2^-64, 2^64) (tests/tools/rcp_exhaustive.hip);
//  * the x / z numerators a = b - o use the lamp's launch-uniform coordinates: k_prepare_launch6
//    writes per-launch node-pair records with that subtraction applied, (min, max) of one axis of
//    one child in a register pair;
//  * ONE record array: pair records [0, P) and 64-byte leaf-triangle records [P, P + T) -- a child
//    reference is its record index, so the fetch address is one shift-add for inner and leaf lanes;
//  * the 175 most visited records of the lamp (two thirds of the node visits) are served from LDS;
//  * leaf visits only on every LEAFP-th trip (lanes standing at a leaf wait);
//  * descend / push / pop as selects instead of nested branches.
//
// Lanes or launches outside the proof conditions (direction component zero, > 1 or < 2^-60, tiny or
// huge origin / scene bounds) run the EXACT instantiation of the step: IEEE divisions and OpenCL's
// select-form min/max, as the reference writes them.
#include "uvrt_traverse.h"

namespace uvrt {

// `exact` is wave-uniform: the record fetch and the descend / push / pop logic are common, only the
// arithmetic of the box and triangle tests differs.
template <bool TOP, int FL>
__device__ __forceinline__ void step6(Lane6& L, const ExtendParams& p, uint32_t stack_base,
                                      const float4* s_top, uint32_t top_pairs, bool leaf_trip, bool exact,
                                      unsigned long long m_act /* lanes holding a ray */)
{
    const uint32_t cur = L.cur;
    const bool is_inner = cur < REF_LEAF_BIT;
    const bool is_leaf = (cur >= REF_LEAF_BIT) & (cur != REF_DONE) & leaf_trip;
    const uint32_t idx = cur & REF_FIRST_MASK;          // record index (inner indices are < 2^27 too)
    // ONE asm block fetches the 64-byte record of every stepping lane -- from the LDS top-of-tree
    // cache or from global memory, chosen by exec masks -- and the lane's stack top, so that both
    // sources write the same registers (hipcc otherwise merges the two branches with v_mov chains).
    v4f w0, w1, w2, w3;
    uint32_t spec_top = REF_DONE;                       // stays REF_DONE when the stack is empty
    // stack entry sp - 1 of this lane (lanes with sp == 0 are masked off); entry sp is 1024 bytes on
    const uint32_t sa = stack_base + ((uint32_t)L.sp << 10);
    {
        // lane masks from single comparisons, combined as 64-bit integers (SALU): a ballot of a
        // compound condition would go through a v_cndmask / v_cmp pair
        const unsigned long long m_in = __builtin_amdgcn_ballot_w64(cur < REF_LEAF_BIT);
        const unsigned long long m_top = TOP ? __builtin_amdgcn_ballot_w64(cur < top_pairs) : 0ull;
        const unsigned long long m_sp = __builtin_amdgcn_ballot_w64(L.sp > 0);
        const unsigned long long m_go = m_in | (leaf_trip ? (m_act & ~m_in) : 0ull);
        const unsigned long long m_glob = m_go & ~m_top;
        const unsigned long long m_stk = m_go & m_sp;
        // LDS copy of record r at byte TOP6_STRIDE * r
        const uint32_t a0 = (uint32_t)(uintptr_t)s_top + cur * TOP6_STRIDE;
        // byte offset of the record: the shift drops the leaf flag and count bits of a reference
        // (record indices are < 2^26: uvrt_capi.hip checks P + T), the base address is scalar
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
                     "s_mov_b64 exec, %[mglob]\n\t"
                     "global_load_dwordx4 %[w0], %[ro], %[rb]\n\t"
                     "global_load_dwordx4 %[w1], %[ro], %[rb] offset:16\n\t"
                     "global_load_dwordx4 %[w2], %[ro], %[rb] offset:32\n\t"
                     "global_load_dwordx4 %[w3], %[ro], %[rb] offset:48\n\t"
                     "s_mov_b64 exec, %[save]\n\t"
                     "s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [st] "+v"(spec_top),
                       [save] "=&s"(save)
                     : [a0] "v"(a0), [sa] "v"(sa), [ro] "v"(roff), [rb] "s"(p.recs),
                       [mtop] "s"(m_top), [mglob] "s"(m_glob), [mstk] "s"(m_stk)
                     : "memory");
    }
    bool need_pop = is_leaf;
    if (is_inner) {
        float d0, d1;
        bool h0, h1;
        if (FL == 2) {        // "shipped flags": t = (b - o) * v_rcp_f32(d) for every lane (there is no other form of it)
            h0 = box_shipped(w0.x, w0.y, w2.x - L.po.x, w2.y - L.po.x, w0.z, w0.w, L.px.y, L.py.y, L.pz.y, L.po.y, d0);
            h1 = box_shipped(w1.x, w1.y, w2.z - L.po.x, w2.w - L.po.x, w1.z, w1.w, L.px.y, L.py.y, L.pz.y, L.po.y, d1);
        } else if (exact) {
            h0 = box_exact(w0.x, w0.y, w2.x - L.po.x, w2.y - L.po.x, w0.z, w0.w, L.px.x, L.py.x, L.pz.x, L.po.y, d0);
            h1 = box_exact(w1.x, w1.y, w2.z - L.po.x, w2.w - L.po.x, w1.z, w1.w, L.px.x, L.py.x, L.pz.x, L.po.y, d1);
        } else {
            v2f x0 = __builtin_shufflevector(w0, w0, 0, 1), z0 = __builtin_shufflevector(w0, w0, 2, 3);
            v2f x1 = __builtin_shufflevector(w1, w1, 0, 1), z1 = __builtin_shufflevector(w1, w1, 2, 3);
            v2f y0 = __builtin_shufflevector(w2, w2, 0, 1), y1 = __builtin_shufflevector(w2, w2, 2, 3);
            slabs6(x0, y0, z0, L.px, L.py, L.pz, L.po);
            h0 = box_fast(x0, y0, z0, L.po.y, d0);
            slabs6(x1, y1, z1, L.px, L.py, L.pz, L.po);
            h1 = box_fast(x1, y1, z1, L.po.y, d1);
        }
        // extend.cl:56-76 with dist = 1e30f for a missed child: nearer first, farther pushed
        // dist1 > dist2 of extend.cl:61 with 1e30f standing for a miss: child 1 first iff it is hit and
        // child 0 is missed or farther (a hit distance is < dist <= 1e30f, so the sentinel never ties)
        const bool sw = h1 & (!h0 | (d0 > d1));
        const uint32_t r0 = __float_as_uint(w3.x), r1 = __float_as_uint(w3.y);
        const uint32_t nearer = sw ? r1 : r0, farther = sw ? r0 : r1;
        if (h0 & h1) {
            if (L.sp < PS6) asm volatile("ds_write_b32 %0, %1 offset:1024" : : "v"(sa), "v"(farther) : "memory");
            else if (L.sp < MAXS6) ovf_ptr(p, stack_base)[L.sp - PS6] = farther;
            else *p.error_flag = 1u;
            L.sp = L.sp < MAXS6 ? L.sp + 1 : L.sp;
        }
        need_pop = !(h0 | h1);
        L.cur = nearer;
    }
    // extend.cl:48-55 -- AFTER the inner-node block (other lanes): the triangles of a leaf with several of them
    // are fetched when the node records' registers are free again
    if (is_leaf) {
        uint32_t count = (cur >> REF_COUNT_SHIFT) & 15u;
        const uint32_t first = idx - (uint32_t)p.npairs;
        if (count == 15u) count = p.scene.leaf_count[first];
        float dist = L.po.y;
        tri6<FL>(p.ox, L.po.x, p.oz, L.px.x, L.py.x, L.pz.x, dist, L.triID,
             make_float4(w0.x, w0.y, w0.z, w0.w), make_float4(w1.x, w1.y, w1.z, w1.w),
             make_float4(w2.x, w2.y, w2.z, w2.w), exact);
        for (uint32_t i = 1; i < count; ++i) {
            const float4* lt = (const float4*)p.recs + ((size_t)idx + i) * 4;
            tri6<FL>(p.ox, L.po.x, p.oz, L.px.x, L.py.x, L.pz.x, dist, L.triID, lt[0], lt[1], lt[2], exact);
        }
        L.po.y = dist;
    }
    if (need_pop) {
        uint32_t popped = spec_top;                        // REF_DONE when the stack is empty
        if (L.sp > PS6) popped = ovf_ptr(p, stack_base)[L.sp - 1 - PS6];
        L.cur = popped;
        L.sp = (int)__builtin_elementwise_sub_sat((uint32_t)L.sp, 1u);
    }
}

// The same step for the common case -- no lane needs the IEEE-division form, no lane's stack has left LDS -- with
// the control flow written as lane masks instead of divergent branches.  Scalar issue is the dearest resource of
// this kernel (one instruction per cycle per CU, shared by 32 waves: 32 extra scalar instructions per trip cost
// 17 % of the launch, profiles/r02/r02_experiments.txt), and hipcc spends ~70 of them per trip on exec bookkeeping
// for `if (inner) {...} if (both hit) {push} if (none hit) {pop}`.  Here the caller hands over the lane masks of
// the trip (one vector comparison each), the box arithmetic runs for ALL lanes (a vector instruction costs the
// same whatever its exec mask; lanes that do not stand at an inner node compute on stale registers and are
// masked out of the results), the hit tests narrow exec themselves (v_cmpx), and descend / push / pop are
// exec-masked instructions of one asm block.
//   m_in: lanes at an inner node, m_leaf: lanes that visit their leaf in this trip, m_top: lanes whose record is
//   in the LDS cache, full: the exec mask of the loop (all 64 lanes)
template <bool TOP, int FL>
__device__ __forceinline__ void step7(Lane6& L, const ExtendParams& p, uint32_t stack_base, uint32_t top_base,
                                      unsigned long long m_in, unsigned long long m_leaf, unsigned long long m_top,
                                      unsigned long long full
#ifdef UVRT_TRIP_STATS
                                      , uint32_t (&clk)[4]
#define UVRT_CLK(i, t0) do { const unsigned long long t1_ = __builtin_readcyclecounter(); clk[i] += (uint32_t)(t1_ - t0); t0 = t1_; } while (0)
#else
#define UVRT_CLK(i, t0) do { } while (0)
#endif
                                      )
{
    const uint32_t cur = L.cur;
    v4f w0, w1, w2, w3;
    uint32_t spec_top;
#ifdef UVRT_TRIP_STATS
    unsigned long long tclk = __builtin_readcyclecounter();
#endif
    const uint32_t sa = stack_base + ((uint32_t)L.sp << 10);
    {
        const unsigned long long m_glob = (m_in | m_leaf) & ~m_top;
        const uint32_t a0 = __umul24(cur, TOP6_STRIDE) + top_base;      // only used by lanes in m_top
        const uint32_t roff = cur << 6;
        // the stack top is read by every lane: entry -1 of a lane's LDS stack is a row that always holds REF_DONE
        asm volatile("ds_read_b32 %[st], %[sa]\n\t"
                     "s_mov_b64 exec, %[mtop]\n\t"
                     "ds_read_b128 %[w0], %[a0]\n\t"
                     "ds_read_b128 %[w1], %[a0] offset:16\n\t"
                     "ds_read_b128 %[w2], %[a0] offset:32\n\t"
                     "ds_read_b128 %[w3], %[a0] offset:48\n\t"
                     "s_mov_b64 exec, %[mglob]\n\t"
                     "global_load_dwordx4 %[w0], %[ro], %[rb]\n\t"
                     "global_load_dwordx4 %[w1], %[ro], %[rb] offset:16\n\t"
                     "global_load_dwordx4 %[w2], %[ro], %[rb] offset:32\n\t"
                     "global_load_dwordx4 %[w3], %[ro], %[rb] offset:48\n\t"
                     "s_mov_b64 exec, %[full]\n\t"
                     "s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [st] "=&v"(spec_top)
                     : [a0] "v"(a0), [sa] "v"(sa), [ro] "v"(roff), [rb] "s"(p.recs), [mtop] "s"(m_top), [mglob] "s"(m_glob),
                       [full] "s"(full)
                     : "memory");
    }
    UVRT_CLK(0, tclk);
    if (m_leaf != 0) {                                       // wave-uniform; m_leaf != 0 means: a leaf trip
        if ((int32_t)cur < -1) {                             // at a leaf (REF_DONE is -1): extend.cl:48-55
            const uint32_t idx = cur & REF_FIRST_MASK;
            uint32_t count = (cur >> REF_COUNT_SHIFT) & 15u;
            const uint32_t first = idx - (uint32_t)p.npairs;
            if (count == 15u) count = p.scene.leaf_count[first];
            float dist = L.po.y;
            tri6<FL>(p.ox, L.po.x, p.oz, L.px.x, L.py.x, L.pz.x, dist, L.triID,
                      make_float4(w0.x, w0.y, w0.z, w0.w), make_float4(w1.x, w1.y, w1.z, w1.w),
                      make_float4(w2.x, w2.y, w2.z, w2.w), false);
            for (uint32_t i = 1; i < count; ++i) {
                const float4* lt = (const float4*)p.recs + ((size_t)idx + i) * 4;
                tri6<FL>(p.ox, L.po.x, p.oz, L.px.x, L.py.x, L.pz.x, dist, L.triID, lt[0], lt[1], lt[2], false);
            }
            L.po.y = dist;
        }
    }
    UVRT_CLK(1, tclk);
    if (m_in != 0) {            // wave-uniform: a trip with no lane at an inner node skips the box arithmetic
        v2f x0 = __builtin_shufflevector(w0, w0, 0, 1), z0 = __builtin_shufflevector(w0, w0, 2, 3);
        v2f x1 = __builtin_shufflevector(w1, w1, 0, 1), z1 = __builtin_shufflevector(w1, w1, 2, 3);
        v2f y0 = __builtin_shufflevector(w2, w2, 0, 1), y1 = __builtin_shufflevector(w2, w2, 2, 3);
        if (FL == 2) {
            slabs6s(x0, y0, z0, L.px, L.py, L.pz, L.po);
            slabs6s(x1, y1, z1, L.px, L.py, L.pz, L.po);
        } else {
            slabs6(x0, y0, z0, L.px, L.py, L.pz, L.po);
            slabs6(x1, y1, z1, L.px, L.py, L.pz, L.po);
        }
        float n0, f0, n1, f1;
        box2_fast(x0, y0, z0, x1, y1, z1, n0, f0, n1, f1);
        // extend.cl:36-38,56-76: hit = tmax >= tmin && tmin < dist && tmax > 0 per child; child 1 first iff it is hit
        // and child 0 is missed or farther; both hit: the farther one is pushed; none hit (or a leaf visited): pop
        unsigned long long h0, h1, t;
        asm volatile("s_mov_b64 exec, %[min]\n\t"
                     "v_cmpx_ge_f32_e64 %[h0], %[f0], %[n0]\n\t"
                     "v_cmpx_lt_f32_e64 %[h0], %[n0], %[dist]\n\t"
                     "v_cmpx_gt_f32_e64 %[h0], %[f0], 0\n\t"            // h0 = exec = inner lanes whose child 0 is hit
                     "s_mov_b64 exec, %[min]\n\t"
                     "v_cmpx_ge_f32_e64 %[h1], %[f1], %[n1]\n\t"
                     "v_cmpx_lt_f32_e64 %[h1], %[n1], %[dist]\n\t"
                     "v_cmpx_gt_f32_e64 %[h1], %[f1], 0\n\t"            // h1 likewise
                     "v_cmp_gt_f32 vcc, %[n0], %[n1]\n\t"               // (under exec = h1)
                     "s_andn2_b64 %[t], %[h1], %[h0]\n\t"
                     "s_or_b64 %[t], %[t], vcc\n\t"                     // t = child 1 first
                     "s_and_b64 exec, %[h0], %[h1]\n\t"                 // both hit: push the farther, sp + 1
                     "v_cndmask_b32 %[n1], %[r1], %[r0], %[t]\n\t"
                     "ds_write_b32 %[sa], %[n1] offset:1024\n\t"
                     "v_add_u32 %[sp], 1, %[sp]\n\t"
                     "s_or_b64 exec, %[h0], %[h1]\n\t"                  // any hit: descend into the nearer
                     "v_cndmask_b32 %[cur], %[r0], %[r1], %[t]\n\t"
                     "s_andn2_b64 %[t], %[min], exec\n\t"
                     "s_or_b64 exec, %[t], %[mleaf]\n\t"                // none hit, or a leaf was visited: pop
                     "v_mov_b32 %[cur], %[st]\n\t"
                     "v_sub_u32 %[sp], %[sp], 1 clamp\n\t"
                     "s_mov_b64 exec, %[full]"
                     : [n1] "+v"(n1), [cur] "+v"(L.cur), [sp] "+v"(L.sp), [h0] "=&s"(h0), [h1] "=&s"(h1), [t] "=&s"(t)
                     : [n0] "v"(n0), [f0] "v"(f0), [f1] "v"(f1), [dist] "v"(L.po.y), [r0] "v"(w3.x), [r1] "v"(w3.y), [sa] "v"(sa),
                       [st] "v"(spec_top), [min] "s"(m_in), [mleaf] "s"(m_leaf), [full] "s"(full)
                     : "vcc", "memory");
    } else {
        // only leaves were visited: pop them
        asm volatile("s_mov_b64 exec, %[mleaf]\n\t"
                     "v_mov_b32 %[cur], %[st]\n\t"
                     "v_sub_u32 %[sp], %[sp], 1 clamp\n\t"
                     "s_mov_b64 exec, %[full]"
                     : [cur] "+v"(L.cur), [sp] "+v"(L.sp)
                     : [st] "v"(spec_top), [mleaf] "s"(m_leaf), [full] "s"(full));
    }
    UVRT_CLK(2, tclk);
}

// ---- the common trips as ONE hand-written instruction stream ------------------------------------------------
// run7 executes common trips (what step7 does for one) back to back until something else has to happen and says
// what: 1 = refill (at most `active_min` lanes hold a ray), 2 = a trip for the general step (a lane needs the
// IEEE-division form or its stack has left LDS, or a leaf of this trip holds more than one triangle).  hipcc's
// code for the same loop spends ~43 scalar and ~75 vector instructions on an inner-node trip, a third of them
// exec / phi bookkeeping between the trip's masks and its record fetch; this stream has 31 and 57, and the
// chain from a trip's descend to the next trip's fetch is 8 vector + 14 scalar instructions.
//
// Registers: the lane state are asm operands; everything else lives in v40-v63 (declared clobbered), five scalar
// pairs m0-m4 and the 32-bit %[code].
//   W0-W3 = v[40:43] v[44:47] v[48:51] v[52:55]   the fetched record (inner: child 0 x,z | child 1 x,z | y | refs)
//   ST v56 (stack top), SA v57 (its address); the record's LDS / memory address sits in v52 / v53 until the last
//   quarter of the record lands there; v[58:63] slab temporaries, then box distances (+ v54).  Leaf lanes (exec =
//   m3) use v47, v51-v55 and v57-v63 for the triangle test: the other lanes' copies of those registers are untouched
//   m0 = inner lanes; m1 = lanes at a leaf, later "child 1 hit"; m2 = lanes with a cached record, later "child 0
//   hit" (and the v_cmpx results of the triangle test); m3 = lanes visiting their leaf; m4 = scratch masks;
//   %[code] = scratch (lane count, leaf-trip flag) until it carries the exit code
// Tried and measured in round 3 (profiles/r03/r03_experiments.txt), both bit-exact, neither kept:
//  * an adaptive leaf rule (leaf trip when >= K lanes stand at a leaf or P trips after the last leaf visit; git cc58a8b):
//    no (P, K) beat the fixed alternation, and its seven extra scalar instructions per trip cost 0.8 %;
//  * a one-dword load past L1 (sc1) of a pushed child's record at the push -- a pushed node IS visited later,
//    extend.cl:77-79 -- into the one dead register of the record window, v55: -10.7 % on the room, -2.9 % on a
//    6 M-triangle soup beyond every cache.
//  * run-ahead of the lanes at cached records while the trip's global loads are in flight (their own copies of the record
//    window under exec = those lanes, repeated while they keep landing on cached records): -4 % on a 6 M-triangle soup, -25 % on
//    a 1 M one, -63 % on the room (profiles/r03/r03_run_ahead.patch): with 28 waves per CU one wave's wait is the others' issue time.
// Arithmetic: slabs / boxes / hit tests are step7's (slabs6, box2_fast, the v_cmpx tail); the triangle test is
// tri6<FL> instruction for instruction (extend.cl:6-27), early returns as v_cmpx narrowing of exec.
// A kernel with this stream must not spill: scratch use costs the launch pipelining 14 % (measured); the general
// step is ordered (inner block before leaf block) so that hipcc's allocation fits in the 64 registers.
#define R7_CROSS_STRICT(dst, ay, bz, az, by) \
    "v_mul_f32 v58, " ay ", " bz "\n\t" "v_mul_f32 v59, " az ", " by "\n\t" "v_sub_f32 " dst ", v58, v59\n\t"
#define R7_CROSS_OCL(dst, ay, bz, az, by) \
    "v_mul_f32 v59, " az ", " by "\n\t" "v_fma_f32 " dst ", " ay ", " bz ", -v59\n\t"
// dot(a, b): strict = (ax*bx + ay*by) + az*bz in source order; ocl = fma(az, bz, fma(ay, by, ax*bx))
#define R7_DOT_STRICT(dst, ax, ay, az, bx, by, bz) \
    "v_mul_f32 v58, " ax ", " bx "\n\t" "v_mul_f32 v59, " ay ", " by "\n\t" "v_add_f32 v58, v58, v59\n\t" \
    "v_mul_f32 v59, " az ", " bz "\n\t" "v_add_f32 " dst ", v58, v59\n\t"
#define R7_DOT_OCL(dst, ax, ay, az, bx, by, bz) \
    "v_mul_f32 v58, " ax ", " bx "\n\t" "v_fma_f32 v58, " ay ", " by ", v58\n\t" "v_fma_f32 " dst ", " az ", " bz ", v58\n\t"
// NEWTON: the refinement of f (flavours 0 / 1: f = RN(1 / a)) or nothing (flavour 2: f = v_rcp_f32(a));
// the early returns (v_cmpx narrows exec to the lanes that go on): flavours 0 / 1 negate the source's return conditions
// (a NaN goes on, as in extend.cl built strictly), flavour 2 tests the continue conditions the no-NaN licence of the
// reference's own build flags makes of them (ref_extend_fast.co: |a| >= 1e-5, 0 <= u, 1 >= u, 0 <= v, 1 >= v + u)
#define R7_NEWTON_EXACT \
    "v_fma_f32 v57, -v55, v60, 1.0\n\t"                                     /* f = RN(1 / a): rcp + one Newton step */ \
    "v_fma_f32 v60, v57, v60, v60\n\t"
#define R7_NEWTON_NONE
#define R7_GO_A_STRICT "v_cmpx_ngt_f32_e32 vcc, 0x3727c5ac, v58\n\t"         /* if (fabs(a) < 1e-5f) return */
#define R7_GO_A_SHIPPED "v_cmpx_le_f32_e32 vcc, 0x3727c5ac, v58\n\t"
#define R7_GO_01_STRICT(x, pre, y) \
    "v_cmpx_nlt_f32_e64 %[m2], " x ", 0\n\t" pre "v_cmpx_ngt_f32_e64 %[m2], " y ", 1.0\n\t"
#define R7_GO_01_SHIPPED(x, pre, y) \
    "v_cmpx_ge_f32_e64 %[m2], " x ", 0\n\t" pre "v_cmpx_le_f32_e64 %[m2], " y ", 1.0\n\t"
#define R7_TRI(CROSS, DOT, NEWTON, GO_A, GO_01)                                                                     \
    /* h = cross(dir, e2) -> v52 v53 v54 */                                                                         \
    CROSS("v52", "%[dy]", "v50", "%[dz]", "v49")                                                                    \
    CROSS("v53", "%[dz]", "v48", "%[dx]", "v50")                                                                    \
    CROSS("v54", "%[dx]", "v49", "%[dy]", "v48")                                                                    \
    DOT("v55", "v44", "v45", "v46", "v52", "v53", "v54")                    /* a = dot(e1, h) */                    \
    "v_and_b32 v58, 0x7fffffff, v55\n\t"                                                                                 \
    GO_A                                                                                                            \
    "v_rcp_f32 v60, v55\n\t"                                                                                        \
    "v_sub_f32 v61, %[ox], v40\n\t"                                         /* s = orig - v0 */                     \
    "v_sub_f32 v62, %[oy], v41\n\t"                                                                                 \
    "v_sub_f32 v63, %[oz], v42\n\t"                                                                                 \
    NEWTON                                                                                                          \
    DOT("v51", "v61", "v62", "v63", "v52", "v53", "v54")                                                            \
    "v_mul_f32 v51, v60, v51\n\t"                                           /* u = f * dot(s, h) */                 \
    GO_01("v51", "", "v51")                                                 /* if (u < 0 || u > 1) return */        \
    CROSS("v52", "v62", "v46", "v63", "v45")                                /* q = cross(s, e1) */                  \
    CROSS("v53", "v63", "v44", "v61", "v46")                                                                        \
    CROSS("v54", "v61", "v45", "v62", "v44")                                                                        \
    DOT("v47", "%[dx]", "%[dy]", "%[dz]", "v52", "v53", "v54")                                                      \
    "v_mul_f32 v47, v60, v47\n\t"                                           /* v = f * dot(dir, q) */               \
    GO_01("v47", "v_add_f32 v55, v51, v47\n\t", "v55")                       /* if (v < 0 || u + v > 1) return */    \
    DOT("v55", "v48", "v49", "v50", "v52", "v53", "v54")                                                            \
    "v_mul_f32 v55, v60, v55\n\t"                                           /* t = f * dot(e2, q) */                \
    "v_cmpx_lt_f32_e32 vcc, 0x38d1b717, v55\n\t"                             /* if (t > 1e-4f && t < dist) */        \
    "v_cmpx_lt_f32_e64 %[m2], v55, %[dist]\n\t"                                                                  \
    "v_mov_b32 %[dist], v55\n\t"                                                                                    \
    "v_mov_b32 %[tri], v43\n\t"

#define R7_CLOBBERS "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63"
// the twelve slab distances of a node pair (extend.cl:31-35), x / z numerators already b - o:
//   exact (flavours 0 / 1): t = RN((b - o) / d) as q0 = a * y; r = fma(-d, q0, a); q = fma(r, y, q0), y = RN(1 / d) -- 2 + 18 packed ops
//   shipped (flavour 2):    t = (b - o) * v_rcp_f32(d), what the reference's own build flags compile -- 2 + 6 packed ops
#define R7_SLABS_EXACT \
        "v_pk_add_f32 v[48:49], v[48:49], %[po] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
        "v_pk_mul_f32 v[58:59], v[40:41], %[px] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_mul_f32 v[60:61], v[42:43], %[pz] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_mul_f32 v[62:63], v[48:49], %[py] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_fma_f32 v[40:41], %[px], v[58:59], v[40:41] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[42:43], %[pz], v[60:61], v[42:43] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[48:49], %[py], v[62:63], v[48:49] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[40:41], v[40:41], %[px], v[58:59] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 v[42:43], v[42:43], %[pz], v[60:61] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 v[48:49], v[48:49], %[py], v[62:63] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_add_f32 v[50:51], v[50:51], %[po] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
        "v_pk_mul_f32 v[58:59], v[44:45], %[px] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_mul_f32 v[60:61], v[46:47], %[pz] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_mul_f32 v[62:63], v[50:51], %[py] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_fma_f32 v[44:45], %[px], v[58:59], v[44:45] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[46:47], %[pz], v[60:61], v[46:47] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[50:51], %[py], v[62:63], v[50:51] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t" \
        "v_pk_fma_f32 v[44:45], v[44:45], %[px], v[58:59] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 v[46:47], v[46:47], %[pz], v[60:61] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t" \
        "v_pk_fma_f32 v[50:51], v[50:51], %[py], v[62:63] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
#define R7_SLABS_SHIPPED \
        "v_pk_add_f32 v[48:49], v[48:49], %[po] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
        "v_pk_add_f32 v[50:51], v[50:51], %[po] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t" \
        "v_pk_mul_f32 v[40:41], v[40:41], %[px] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_mul_f32 v[42:43], v[42:43], %[pz] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_mul_f32 v[44:45], v[44:45], %[px] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_mul_f32 v[46:47], v[46:47], %[pz] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_mul_f32 v[48:49], v[48:49], %[py] op_sel:[0,1] op_sel_hi:[1,1]\n\t" \
        "v_pk_mul_f32 v[50:51], v[50:51], %[py] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
#define R7_BODY(TRI, SLABS) \
        "1:\n\t" \
        "v_cmp_lt_i32_e64 %[m0], -1, %[cur]\n\t" \
        "v_cmp_gt_i32_e64 %[m1], -1, %[cur]\n\t" \
        "v_cmp_lt_u32_e64 vcc, 7, %[sp]\n\t" \
        "s_or_b64 %[m4], %[m0], %[m1]\n\t" \
        "s_bcnt1_i32_b64 %[code], %[m4]\n\t" \
        "s_cmp_le_u32 %[code], %[amin]\n\t" \
        "s_cbranch_scc1 7f\n\t" \
        "s_or_b64 %[m4], vcc, %[spec]\n\t" \
        "s_cbranch_scc1 8f\n\t" \
        "v_cmp_gt_u32_e64 %[m2], %[tp], %[cur]\n\t" \
        "s_cmp_eq_u64 %[m0], 0\n\t" \
        "s_cselect_b32 %[code], -1, %[km]\n\t" \
        "s_cmp_lg_u32 %[code], 0\n\t" \
        "s_cselect_b64 %[m3], %[m1], 0\n\t" \
        "s_cmp_lg_u64 %[m3], 0\n\t" \
        "s_cbranch_scc0 3f\n\t" \
        "s_mov_b64 exec, %[m3]\n\t" \
        "v_bfe_u32 v58, %[cur], 27, 4\n\t" \
        "v_cmp_ne_u32_e64 %[m4], 1, v58\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        "s_cmp_lg_u64 %[m4], 0\n\t" \
        "s_cbranch_scc1 8f\n\t" \
        "3:\n\t" \
        "s_not_b32 %[km], %[km]\n\t" \
        "v_lshl_add_u32 v57, %[sp], 10, %[sb]\n\t" \
        "v_mul_u32_u24 v52, 0x40, %[cur]\n\t" \
        "v_add_u32 v52, %[tb], v52\n\t" \
        "v_lshlrev_b32 v53, 6, %[cur]\n\t" \
        "s_or_b64 %[m4], %[m0], %[m3]\n\t" \
        "s_andn2_b64 %[m4], %[m4], %[m2]\n\t" \
        "ds_read_b32 v56, v57\n\t" \
        "s_mov_b64 exec, %[m2]\n\t" \
        "ds_read_b128 v[40:43], v52\n\t" \
        "ds_read_b128 v[44:47], v52 offset:16\n\t" \
        "ds_read_b128 v[48:51], v52 offset:32\n\t" \
        "ds_read_b128 v[52:55], v52 offset:48\n\t" \
        "s_mov_b64 exec, %[m4]\n\t" \
        "global_load_dwordx4 v[40:43], v53, %[rb]\n\t" \
        "global_load_dwordx4 v[44:47], v53, %[rb] offset:16\n\t" \
        "global_load_dwordx4 v[48:51], v53, %[rb] offset:32\n\t" \
        "global_load_dwordx4 v[52:55], v53, %[rb] offset:48\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t" \
        "s_cmp_eq_u64 %[m3], 0\n\t" \
        "s_cbranch_scc1 4f\n\t" \
        "s_mov_b64 exec, %[m3]\n\t" \
        TRI \
        "s_mov_b64 exec, -1\n\t" \
        "4:\n\t" \
        "s_cmp_eq_u64 %[m0], 0\n\t" \
        "s_cbranch_scc1 5f\n\t" \
        "s_mov_b64 exec, %[m0]\n\t"   /* the box arithmetic under the inner lanes' mask: same issue cost, less switching power */ \
        SLABS \
        "v_min_f32 v58, v40, v41\n\t" \
        "v_min_f32 v59, v48, v49\n\t" \
        "v_min_f32 v60, v42, v43\n\t" \
        "v_max3_f32 v61, v58, v59, v60\n\t" \
        "v_max_f32 v58, v40, v41\n\t" \
        "v_max_f32 v59, v48, v49\n\t" \
        "v_max_f32 v60, v42, v43\n\t" \
        "v_min3_f32 v62, v58, v59, v60\n\t" \
        "v_min_f32 v58, v44, v45\n\t" \
        "v_min_f32 v59, v50, v51\n\t" \
        "v_min_f32 v60, v46, v47\n\t" \
        "v_max3_f32 v63, v58, v59, v60\n\t" \
        "v_max_f32 v58, v44, v45\n\t" \
        "v_max_f32 v59, v50, v51\n\t" \
        "v_max_f32 v60, v46, v47\n\t" \
        "v_min3_f32 v54, v58, v59, v60\n\t" \
        "v_cmpx_ge_f32_e64 %[m2], v62, v61\n\t" \
        "v_cmpx_lt_f32_e64 %[m2], v61, %[dist]\n\t" \
        "v_cmpx_gt_f32_e64 %[m2], v62, 0\n\t" \
        "s_mov_b64 exec, %[m0]\n\t" \
        "v_cmpx_ge_f32_e64 %[m1], v54, v63\n\t" \
        "v_cmpx_lt_f32_e64 %[m1], v63, %[dist]\n\t" \
        "v_cmpx_gt_f32_e64 %[m1], v54, 0\n\t" \
        "v_cmp_gt_f32 vcc, v61, v63\n\t" \
        "s_andn2_b64 %[m4], %[m1], %[m2]\n\t" \
        "s_or_b64 %[m4], %[m4], vcc\n\t" \
        "s_and_b64 exec, %[m2], %[m1]\n\t" \
        "v_cndmask_b32 v63, v53, v52, %[m4]\n\t" \
        "ds_write_b32 v57, v63 offset:1024\n\t" \
        "v_add_u32 %[sp], 1, %[sp]\n\t" \
        "s_or_b64 exec, %[m2], %[m1]\n\t" \
        "v_cndmask_b32 %[cur], v52, v53, %[m4]\n\t" \
        "s_andn2_b64 %[m4], %[m0], exec\n\t" \
        "s_or_b64 exec, %[m4], %[m3]\n\t" \
        "v_mov_b32 %[cur], v56\n\t" \
        "v_sub_u32 %[sp], %[sp], 1 clamp\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        "s_branch 1b\n\t" \
        "5:\n\t" \
        "s_mov_b64 exec, %[m3]\n\t" \
        "v_mov_b32 %[cur], v56\n\t" \
        "v_sub_u32 %[sp], %[sp], 1 clamp\n\t" \
        "s_mov_b64 exec, -1\n\t" \
        "s_branch 1b\n\t" \
        "7:\n\t" \
        "s_mov_b32 %[code], 1\n\t" \
        "s_branch 9f\n\t" \
        "8:\n\t" \
        "s_mov_b32 %[code], 2\n\t" \
        "9:"

#define R7_OPERANDS \
        : [cur] "+v"(L.cur), [sp] "+v"(L.sp), [dist] "+v"(dist), [tri] "+v"(L.triID), [km] "+s"(km), [code] "=&s"(code), \
          [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4) \
        : [px] "v"(L.px), [py] "v"(L.py), [pz] "v"(L.pz), [po] "v"(L.po), [dx] "v"(L.px.x), [dy] "v"(L.py.x), [dz] "v"(L.pz.x), \
          [oy] "v"(oy), [sb] "v"(stack_base), [tb] "s"(__builtin_amdgcn_readfirstlane(top_base)), [rb] "s"(p.recs), [spec] "s"(special_mask), \
          [tp] "s"(top_pairs), [amin] "s"(active_min), [ox] "s"(p.ox), [oz] "s"(p.oz) \
        : "memory", "scc", "vcc", R7_CLOBBERS

template <int FL, int LEAFP>
__device__ __forceinline__ int run7(Lane6& L, const ExtendParams& p, uint32_t stack_base, uint32_t top_base,
                                    uint32_t top_pairs, unsigned long long special_mask, int& km,
                                    unsigned long long full, int active_min)
{
    static_assert(LEAFP == 2, "run7 visits leaves in every second trip");
    static_assert(TOP6_STRIDE == 0x40, "run7 multiplies by the literal stride of the LDS cache");
    int code;
    unsigned long long m0, m1, m2, m3, m4;   // scalar temporaries of the stream: lane masks
    // {d, 1/d} per axis, {origin y, dist}: the scalar halves are separate operands (an asm operand has no
    // sub-register syntax), tied to the same registers as the pairs by construction of Lane6
    float dist = L.po.y;
    const float oy = L.po.x;
    static_assert(PS6 == 8, "run7 compares the stack pointer with the literal PS6 - 1");
    (void)full;      // the loop's exec mask is all 64 lanes (full workgroups): the stream writes it as the literal -1, one register pair less
    if constexpr (FL == 2)
        asm volatile(R7_BODY(R7_TRI(R7_CROSS_OCL, R7_DOT_OCL, R7_NEWTON_NONE, R7_GO_A_SHIPPED, R7_GO_01_SHIPPED), R7_SLABS_SHIPPED) R7_OPERANDS);
    else if constexpr (FL == 1)
        asm volatile(R7_BODY(R7_TRI(R7_CROSS_OCL, R7_DOT_OCL, R7_NEWTON_EXACT, R7_GO_A_STRICT, R7_GO_01_STRICT), R7_SLABS_EXACT) R7_OPERANDS);
    else
        asm volatile(R7_BODY(R7_TRI(R7_CROSS_STRICT, R7_DOT_STRICT, R7_NEWTON_EXACT, R7_GO_A_STRICT, R7_GO_01_STRICT), R7_SLABS_EXACT) R7_OPERANDS);
    L.po.y = dist;
    return code;
}

// A ray outside the proof conditions of the packed exact division (a direction component zero, NaN, > 1 or
// < 2^-60; an origin height that is tiny but not zero, or huge): it runs the IEEE-division form of the step.
// Range tests on the bit patterns: |x| in [lo, hi]  <=>  bits(|x|) - bits(lo) <= bits(hi) - bits(lo) as unsigned.
__device__ __forceinline__ bool outside_proof_conditions(float4 rec)
{
    const uint32_t lo = 0x21800000u /* 2^-60 */, one = 0x3F800000u;
    const uint32_t ux = (__float_as_uint(rec.x) & 0x7FFFFFFFu) - lo, uy = (__float_as_uint(rec.y) & 0x7FFFFFFFu) - lo,
                   uz = (__float_as_uint(rec.z) & 0x7FFFFFFFu) - lo;
    const uint32_t worst = max(max(ux, uy), uz);
    const uint32_t uo = __float_as_uint(rec.w) & 0x7FFFFFFFu;                   // |origin y|
    const uint32_t ylo = 0x0D800000u /* 2^-100 = 7.888609e-31f */, yhi = 0x4E6E6B28u /* 1e9f */;
    return worst > one - lo || (uo != 0u && uo - ylo > yhi - ylo);
}

// The drain of a wave's last rays (its share of the launch is handed out, fewer and fewer lanes hold a ray, every trip still costs
// a full trip) is the dearest part of a launch: 25 % of the trips of a 2 M-ray launch.  Once a wave is down to MERGE6_AT rays it
// stops and waits for the other three waves of its workgroup to get there; then all four write their rays into LDS and wave 0 goes
// on with all of them (<= 64) while the others leave: one wave's trips instead of four for the rest of the drain.
//   * a ray keeps the LDS stack rows (and the overflow rows) of the lane that started it: `stack_base` travels with the ray;
//   * the exchange area is the tail of the record cache (nobody traverses between the two barriers; afterwards wave 0 treats
//     only the first TOP6_KEEP records as cached); the four counts sit in the waves' own stack sentinels for the moment;
//   * every wave passes here exactly once (any wave's drain ends at zero rays): the barriers match by construction.
// Which lane finishes a ray changes nothing it deposits.  Returns whether this wave goes on.
#ifndef UVRT_MERGE6_AT
#define UVRT_MERGE6_AT 16
#endif
constexpr uint32_t MERGE6_AT = UVRT_MERGE6_AT, MERGE6_FIELDS = 14;
static_assert(4 * MERGE6_AT <= 64, "the rays of four waves must fit one");
constexpr uint32_t TOP6_KEEP = TOP6_MAX + 1 - (MERGE6_FIELDS * 64 * 4 + TOP6_STRIDE - 1) / TOP6_STRIDE;      // 120 records stay cached
// ints from a wave's replica to the plane of a lane's ray (bit 31 of that register: the ray needs the exact step, see k_extend6)
constexpr uint32_t PLANE_OFF6 = 0x7FFFFFFFu, SPECIAL6 = 0x80000000u;
template <bool RECORD>
__device__ __forceinline__ bool merge6(Lane6& L, const ExtendParams& p, int32_t* my_counts, uint32_t& plane_off, uint32_t& stack_base,
                                       uint32_t& slot, bool& live, uint32_t* row0, uint32_t* xch)
{
    // (few scalars at a time: the kernel's long-lived ones leave little room, and what does not fit is spilled for the whole loop)
    if (L.cur == REF_DONE) {        // results of the rays these lanes finished since the last refill (extend.cl:94-98)
        if (RECORD && live && p.hits) {
            const uint32_t li = p.order ? p.order[slot] : slot;
            p.hits[li] = make_uint2(__float_as_uint(L.po.y), L.triID);
        }
        if (L.po.y != 1e30f) atomicAdd(&my_counts[(plane_off & PLANE_OFF6) + L.triID], 1);
    }
    // the wave's number in its workgroup comes from the lane's own stack rows (not yet anybody else's; threadIdx.x is not kept)
    {
        const uint32_t cnt = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(L.cur != REF_DONE));
        if (lane_now() == 0u) row0[(stack_base >> 2) & 192u] = cnt;
    }
    __syncthreads();
    if (L.cur != REF_DONE) {
        // slot = rays of the waves before this one + the lane's rank among the wave's rays (exec = those lanes here)
        const uint32_t wv = (stack_base >> 8) & 3u;
        uint32_t s = lane_rank_in_exec();
        if (wv > 0u) s += row0[0];
        if (wv > 1u) s += row0[64];
        if (wv > 2u) s += row0[128];
        uint32_t* x = xch + s;
        x[0 * 64] = __float_as_uint(L.px.x); x[1 * 64] = __float_as_uint(L.px.y);
        x[2 * 64] = __float_as_uint(L.py.x); x[3 * 64] = __float_as_uint(L.py.y);
        x[4 * 64] = __float_as_uint(L.pz.x); x[5 * 64] = __float_as_uint(L.pz.y);
        x[6 * 64] = __float_as_uint(L.po.x); x[7 * 64] = __float_as_uint(L.po.y);
        x[8 * 64] = L.triID; x[9 * 64] = L.cur; x[10 * 64] = (uint32_t)L.sp; x[11 * 64] = plane_off; x[12 * 64] = stack_base;
        if (RECORD) x[13 * 64] = slot;
    }
    __syncthreads();
    // this wave's rays are in the exchange area
    const bool goes_on = __builtin_amdgcn_readfirstlane((int)((stack_base >> 8) & 3u)) == 0;
    set_in_place(L.cur, REF_DONE);
    L.po.y = 1e30f;
    live = false;
    if (!goes_on) return false;
    const uint32_t lane = lane_now();
    const uint32_t total = row0[0] + row0[64] + row0[128] + row0[192];
    if (lane < 4u) row0[lane * 64u] = REF_DONE;          // the count words are stack sentinels again
    if (lane < total) {
        const uint32_t* x = xch + lane;
        set_in_place(L.px, __uint_as_float(x[0 * 64]), __uint_as_float(x[1 * 64]));
        set_in_place(L.py, __uint_as_float(x[2 * 64]), __uint_as_float(x[3 * 64]));
        set_in_place(L.pz, __uint_as_float(x[4 * 64]), __uint_as_float(x[5 * 64]));
        set_in_place(L.po, __uint_as_float(x[6 * 64]), __uint_as_float(x[7 * 64]));
        set_in_place(L.triID, x[8 * 64]);
        set_in_place(L.sp, (int)x[10 * 64]);
        set_in_place(plane_off, x[11 * 64]);
        set_in_place(stack_base, x[12 * 64]);
        if (RECORD) { slot = x[13 * 64]; live = true; }
        set_in_place(L.cur, x[9 * 64]);
    }
    return true;
}

template <int LEAFP, bool RECORD, bool TOP, int FL>
__global__ __launch_bounds__(256, 8) void k_extend6(ExtendParams p)
{
    __shared__ __attribute__((aligned(1024))) uint32_t s_stack[PS6 + 1][256];   // 9 KB: row 0 always holds REF_DONE ("entry -1"), the stack proper follows
    __shared__ float4 s_top[TOP ? (TOP6_MAX + 1) * (TOP6_STRIDE / 16) : 4];          // 11 KB
    uint32_t top_pairs = TOP ? (p.top_pairs < TOP6_MAX ? p.top_pairs : TOP6_MAX) : 0u;      // (fewer after the drain merge)
    if (TOP) {
        const float4* src = (const float4*)p.recs;
        for (uint32_t i = threadIdx.x; i < top_pairs * 4u; i += 256u) {
            const uint32_t rec = i >> 2;
            s_top[rec * (TOP6_STRIDE / 16u) + (i & 3u)] = src[i];
        }
        __syncthreads();
    }
    s_stack[0][threadIdx.x] = REF_DONE;                               // what a pop from an empty stack yields
    // LDS byte address of this lane's stack entry -1 (entry e at + 1024 (e + 1))
    uint32_t stack_base = (uint32_t)(uintptr_t)&s_stack[0][threadIdx.x];       // (of the ray's first lane: merge6)
    Lane6 L;
    L.px = L.py = L.pz = (v2f){1.f, 1.f};
    L.po = (v2f){0.f, 1e30f};      // dist == 1e30f <=> nothing to deposit
    L.triID = 0;
    L.cur = REF_DONE;
    L.sp = 0;
    uint32_t slot = 0;
    bool live = false;             // RECORD: holds a ray whose (dist, triID) has not been written yet
    int32_t* const my_counts = p.counts + (int64_t)(blockIdx.x % (unsigned)p.count_replicas) * p.count_stride;
    const uint32_t root6 = (p.perm && p.root_ref6 < REF_LEAF_BIT) ? p.perm[p.root_ref6] : p.root_ref6;
    // ints from my_counts to the plane of the ray this lane holds; bit 31: the ray needs the EXACT step (outside the proof
    // conditions of the packed division) -- the wave's mask of such lanes is a ballot of that bit where it is needed, not a
    // register pair carried through the loop (the kernel runs at its scalar-register budget)
    uint32_t plane_off = 0;
    const float plane_inv = p.plane_inv;

    // wave w traces the 64-ray batches w, w + W, w + 2W, ...: static ownership, no atomics, and batches
    // dealt round-robin so that ordered rays stay load-balanced
    const uint32_t wave = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t W = gridDim.x * 4u;
    uint32_t cursor = 0;
    const uint32_t chunk_end = p.chunk;
    const uint32_t n32 = (uint32_t)p.n;
    uint32_t trip = 0;
    unsigned long long km = ~0ull;          // all ones in a trip that visits leaves (every LEAFP-th)
    unsigned long long full;                // exec of the loop: all 64 lanes (the launch uses full workgroups)
    asm volatile("s_mov_b64 %0, exec" : "=s"(full));
    const uint32_t top_base = (uint32_t)(uintptr_t)s_top;
    // refill when this many lanes are idle; once the wave's sequence is exhausted only the all-idle exit is left
    const int refill_c = p.refill_min;      // 1..64 (launch_extend6)
    int refill_at = refill_c;
    // (cursor == MERGED6 once the wave has been through the workgroup's drain merge (merge6), or from the end of its share on when
    //  the launch does without: the state costs no register of its own)
    constexpr uint32_t MERGED6 = 0xFFFFFFFFu;
#ifdef UVRT_TRIP_STATS     // developer build (tests/tools/trip_stats.sh): where the trips' lanes go
    uint32_t st_trips = 0, st_in = 0, st_leaf = 0, st_wait = 0, st_idle = 0, st_leaftrips = 0, st_drain = 0, st_slow = 0,
             st_refills = 0, st_top = 0, st_d9 = 0, st_d10 = 0, st_d11 = 0, st_d12 = 0;
    // the trips by the kind of work the product's instruction stream (run7) does for them -- the census that prices the kernel's
    // VALU issue cycles (tests/tools/stream_census.py): stream trips with / without the box block and with / without the
    // triangle block, trips the stream hands to the general step (by its arithmetic form), stream exits
    uint32_t st_s_in = 0, st_s_leaf = 0, st_s_both = 0, st_g_exact = 0, st_g_other = 0, st_tris = 0;
    uint32_t clk4 = 0;                // cycles in the general step
    uint32_t clk[4] = {0, 0, 0, 0};   // cycles in: record fetch (issue to data), leaf tests, box tests + descend, refill
    const unsigned long long t_begin = __builtin_readcyclecounter();
#endif

#ifdef UVRT_TRIP_STATS
    constexpr bool asm_trips = false;       // the statistics live in the C++ form of the loop
#else
    constexpr bool asm_trips = TOP && LEAFP == 2;
#endif
    if constexpr (asm_trips) {
        int kflag = -1;                         // -1 in a trip that visits leaves (every second one)
        for (;;) {
            // common trips back to back (run7), until lanes want new rays (1) or a trip needs the general step (2)
            // (scalars that pass through the asm statement are re-declared uniform: hipcc otherwise treats them, and
            // every loop-carried scalar whose update depends on them, as divergent and keeps them in vector registers)
            // (two call sites: a select between the two thresholds would drag `cursor` into a vector register)
            int why;
            const unsigned long long special_mask = FL == 2 ? 0ull : __builtin_amdgcn_ballot_w64((int32_t)plane_off < 0);
            // rays at which the stream comes back: enough idle lanes for a refill; the wave's share handed out: MERGE6_AT for the
            // drain merge, afterwards none.  ONE call site (each one gets its own registers for the lane state, and the loop then
            // copies it at every join), so the threshold is a value -- computed with scalar instructions spelled out: a C++ select
            // on `cursor` drags it into a vector register.
            int thr;
            asm("s_cmp_eq_u32 %[c], -1\n\t"
                "s_cselect_b32 %[t], 0, %[k]\n\t"
                "s_cmp_lt_u32 %[c], %[e]\n\t"
                "s_cselect_b32 %[t], %[a], %[t]"
                : [t] "=&s"(thr) : [c] "s"(cursor), [e] "s"(chunk_end), [a] "s"(64 - refill_c), [k] "n"(TOP ? MERGE6_AT : 0u) : "scc");
            why = run7<FL, LEAFP>(L, p, stack_base, top_base, top_pairs, special_mask, kflag, full, thr);
            why = __builtin_amdgcn_readfirstlane(why);
            kflag = __builtin_amdgcn_readfirstlane(kflag);
            if (why == 1) {
                const unsigned long long idle_mask = __builtin_amdgcn_ballot_w64(L.cur == REF_DONE);
                const int nidle = __popcll(idle_mask);
                if (cursor < chunk_end) {
                    bool spec = false;
                    if (L.cur == REF_DONE) {
                        // results of the rays these lanes finished since the last refill (extend.cl:94-98)
                        if (RECORD && live && p.hits) {
                            const uint32_t li = p.order ? p.order[slot] : slot;
                            p.hits[li] = make_uint2(__float_as_uint(L.po.y), L.triID);
                        }
                        if (L.po.y != 1e30f) atomicAdd(&my_counts[(plane_off & PLANE_OFF6) + L.triID], 1);
                        live = false;
                        L.po.y = 1e30f;
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                              __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                        const uint32_t v = cursor + rank;
                        const uint32_t gb = (v >> 6) * W + wave;                 // global 64-slot batch
                        const uint32_t my = gb * 64u + (v & 63u);
                        // plane (= launch of a batched trace) of the batch: gb / plane_batches, exact after one
                        // correction step (gb < 2^24 is exact in f32, the rounded reciprocal is off by < 1)
                        uint32_t pl = 0;
                        int32_t within = (int32_t)gb;
                        if (p.plane_stride != 0) {               // wave-uniform: a launch of its own is one plane
                            pl = (uint32_t)((float)gb * plane_inv);
                            within = (int32_t)(gb - pl * p.plane_batches);
                            if (within < 0) { --pl; within += (int32_t)p.plane_batches; }
                            else if ((uint32_t)within >= p.plane_batches) { ++pl; within -= (int32_t)p.plane_batches; }
                        }
                        if (v < chunk_end && my < n32 && (uint32_t)within * 64u + (v & 63u) < p.plane_n) {
                            const float4 rec = p.rays[my];
                            // y = RN32(1/d) (rcp_exact: exact for 2^-64 <= |d| < 2^64; other lanes are `spec`
                            // and never use y); flavour 2: y = v_rcp_f32(d), used by every lane
                            set_in_place(L.px, rec.x, FL == 2 ? rcp_raw(rec.x) : rcp_exact(rec.x));
                            set_in_place(L.py, rec.y, FL == 2 ? rcp_raw(rec.y) : rcp_exact(rec.y));
                            set_in_place(L.pz, rec.z, FL == 2 ? rcp_raw(rec.z) : rcp_exact(rec.z));
                            set_in_place(L.po, rec.w, 1e30f);       // generate.cl:34-35
                            set_in_place(L.triID, 0u);
                            if (RECORD) { slot = my; live = true; }
                            set_in_place(L.sp, 0);
                            set_in_place(L.cur, root6);
                            spec = FL != 2 && (outside_proof_conditions(rec) || p.force_exact != 0);
                            set_in_place(plane_off, pl * p.plane_stride | (spec ? SPECIAL6 : 0u));
                        }
                    }
                    cursor += (uint32_t)nidle;
                    if (cursor >= chunk_end) { refill_at = 64; if (!TOP || p.drain_merge == 0) cursor = MERGED6; }
                } else if (TOP && cursor != MERGED6) {
                    cursor = MERGED6;
                    const bool go_on = merge6<RECORD>(L, p, my_counts, plane_off, stack_base, slot, live, &s_stack[0][0],
                                                      (uint32_t*)&s_top[TOP ? TOP6_KEEP * (TOP6_STRIDE / 16u) : 0u]);
                    if (!go_on) break;
                    top_pairs = top_pairs < TOP6_KEEP ? top_pairs : TOP6_KEEP;
                    continue;
                }
                // (a wave leaves only after the merge: the other waves of its workgroup count on its count)
                if (cursor == MERGED6 && __builtin_amdgcn_ballot_w64(L.cur != REF_DONE) == 0) break;
                continue;
            }
            const unsigned long long m_in = __builtin_amdgcn_ballot_w64((int32_t)L.cur >= 0);
            const unsigned long long m_lf = __builtin_amdgcn_ballot_w64((int32_t)L.cur < -1);
            const bool leaf_trip = m_in == 0 || kflag != 0;
            kflag = ~kflag;
            // (two specialisations of the general step -- hipcc's register allocation for the one with both
            // arithmetic forms does not fit beside the registers run7 reserves)
            if ((special_mask & (m_in | m_lf)) != 0) step6<TOP, FL>(L, p, stack_base, s_top, top_pairs, leaf_trip, true, m_in | m_lf);
            else step6<TOP, FL>(L, p, stack_base, s_top, top_pairs, leaf_trip, false, m_in | m_lf);
        }
    } else
    for (;;) {
        const unsigned long long special_mask = FL == 2 ? 0ull : __builtin_amdgcn_ballot_w64((int32_t)plane_off < 0);
        const unsigned long long idle_mask = __builtin_amdgcn_ballot_w64(L.cur == REF_DONE);
        const int nidle = __popcll(idle_mask);
        if (TOP && cursor >= chunk_end && cursor != MERGED6 && 64 - nidle <= (int)MERGE6_AT) {
            cursor = MERGED6;
            const bool go_on = merge6<RECORD>(L, p, my_counts, plane_off, stack_base, slot, live, &s_stack[0][0],
                                              (uint32_t*)&s_top[TOP ? TOP6_KEEP * (TOP6_STRIDE / 16u) : 0u]);
            if (!go_on) break;
            top_pairs = top_pairs < TOP6_KEEP ? top_pairs : TOP6_KEEP;
            continue;
        }
        if (nidle >= refill_at) {
#ifdef UVRT_TRIP_STATS
            unsigned long long tclk = __builtin_readcyclecounter();
#endif
            if (cursor < chunk_end) {
                bool spec = false;
                if (L.cur == REF_DONE) {
                    // results of the rays these lanes finished since the last refill (extend.cl:94-98)
                    if (RECORD && live && p.hits) {
                        const uint32_t li = p.order ? p.order[slot] : slot;
                        p.hits[li] = make_uint2(__float_as_uint(L.po.y), L.triID);
                    }
                    if (L.po.y != 1e30f) atomicAdd(&my_counts[(plane_off & PLANE_OFF6) + L.triID], 1);
                    live = false;
                    L.po.y = 1e30f;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32),
                                          __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
                    const uint32_t v = cursor + rank;
                    const uint32_t gb = (v >> 6) * W + wave;                 // global 64-slot batch
                    const uint32_t my = gb * 64u + (v & 63u);
                    // plane (= launch of a batched trace) of the batch: gb / plane_batches, exact after one
                    // correction step (gb < 2^24 is exact in f32, the rounded reciprocal is off by < 1)
                    uint32_t pl = 0;
                    int32_t within = (int32_t)gb;
                    if (p.plane_stride != 0) {               // wave-uniform: a launch of its own is one plane
                        pl = (uint32_t)((float)gb * plane_inv);
                        within = (int32_t)(gb - pl * p.plane_batches);
                        if (within < 0) { --pl; within += (int32_t)p.plane_batches; }
                        else if ((uint32_t)within >= p.plane_batches) { ++pl; within -= (int32_t)p.plane_batches; }
                    }
                    if (v < chunk_end && my < n32 && (uint32_t)within * 64u + (v & 63u) < p.plane_n) {
                        const float4 rec = p.rays[my];
                        // y = RN32(1/d) (rcp_exact: exact for 2^-64 <= |d| < 2^64; other lanes are `spec`
                        // and never use y); flavour 2: y = v_rcp_f32(d), used by every lane
                        set_in_place(L.px, rec.x, FL == 2 ? rcp_raw(rec.x) : rcp_exact(rec.x));
                        set_in_place(L.py, rec.y, FL == 2 ? rcp_raw(rec.y) : rcp_exact(rec.y));
                        set_in_place(L.pz, rec.z, FL == 2 ? rcp_raw(rec.z) : rcp_exact(rec.z));
                        set_in_place(L.po, rec.w, 1e30f);       // generate.cl:34-35
                        set_in_place(L.triID, 0u);
                        if (RECORD) { slot = my; live = true; }
                        set_in_place(L.sp, 0);
                        set_in_place(L.cur, root6);
                        spec = FL != 2 && (outside_proof_conditions(rec) || p.force_exact != 0);
                        set_in_place(plane_off, pl * p.plane_stride | (spec ? SPECIAL6 : 0u));
                    }
                }
                cursor += (uint32_t)nidle;
                if (cursor >= chunk_end) { refill_at = 64; if (!TOP || p.drain_merge == 0) cursor = MERGED6; }
#ifdef UVRT_TRIP_STATS
                ++st_refills;
                { const unsigned long long t1_ = __builtin_readcyclecounter(); clk[3] += (uint32_t)(t1_ - tclk); }
#endif
            }
            if (__builtin_amdgcn_ballot_w64(L.cur != REF_DONE) == 0) {
                if (cursor == MERGED6) break;
                continue;
            }
        }
        // lane masks of the trip, one vector comparison each: a reference is an inner record index (>= 0 as a signed
        // number), REF_DONE (-1) or a leaf (< -1)
        const unsigned long long m_in = __builtin_amdgcn_ballot_w64((int32_t)L.cur >= 0);
        const unsigned long long m_lf = __builtin_amdgcn_ballot_w64((int32_t)L.cur < -1);
        const unsigned long long m_top = TOP ? __builtin_amdgcn_ballot_w64(L.cur < top_pairs) : 0ull;
        const unsigned long long m_deep = __builtin_amdgcn_ballot_w64(L.sp >= PS6);
        // leaves are visited in every LEAFP-th trip, and in any trip that has no lane at an inner node
        unsigned long long kme = km;
        if (LEAFP > 1) {
            kme = m_in == 0 ? ~0ull : km;
            if (LEAFP == 2) km = ~km;
            else { ++trip; if (trip == (uint32_t)LEAFP) trip = 0; km = trip == 0 ? ~0ull : 0ull; }
        }
        // the common case in its lane-mask form; a trip with a lane that needs IEEE divisions (the bit of a
        // finished lane stays set until the next refill: the exact form is right for every lane) or whose stack
        // has left LDS takes the general step
#ifdef UVRT_TRIP_STATS
        ++st_trips;
#if UVRT_TRIP_STATS >= 2
        st_in += __popcll(m_in); st_leaf += __popcll(m_lf & kme); st_wait += __popcll(m_lf & ~kme);
        st_idle += 64 - __popcll(m_in | m_lf); st_leaftrips += (m_lf & kme) != 0; st_drain += refill_at == 64 && cursor >= chunk_end;
        st_top += __popcll(m_top);
        st_d9 += __builtin_amdgcn_ballot_w64(L.sp >= 9) != 0; st_d10 += __builtin_amdgcn_ballot_w64(L.sp >= 10) != 0;
        st_d11 += __builtin_amdgcn_ballot_w64(L.sp >= 11) != 0; st_d12 += __builtin_amdgcn_ballot_w64(L.sp >= 12) != 0;
#endif
#endif
#ifdef UVRT_TRIP_STATS
        {
            // run7 leaves to the general step for special lanes, stacks beyond LDS, and leaf trips that visit a leaf with several
            // triangles; everything else is a stream trip
            const unsigned long long m_vis = m_lf & kme;
            const bool multi = __builtin_amdgcn_ballot_w64(((m_vis >> (threadIdx.x & 63)) & 1ull) && ((L.cur >> REF_COUNT_SHIFT) & 15u) != 1u) != 0;
            if ((special_mask | m_deep) != 0 || multi) {
                if ((special_mask & (m_in | m_lf)) != 0) ++st_g_exact; else ++st_g_other;
            } else {
                if (m_in != 0 && m_vis != 0) ++st_s_both; else if (m_in != 0) ++st_s_in; else ++st_s_leaf;
            }
            st_tris += (uint32_t)__popcll(m_vis);
        }
#endif
        if ((special_mask | m_deep) != 0) {
#ifdef UVRT_TRIP_STATS
            const unsigned long long t0_ = __builtin_readcyclecounter();
#endif
            step6<TOP, FL>(L, p, stack_base, s_top, top_pairs, kme != 0, (special_mask & (m_in | m_lf)) != 0, m_in | m_lf);
#ifdef UVRT_TRIP_STATS
            clk4 += (uint32_t)(__builtin_readcyclecounter() - t0_);
            ++st_slow;
#endif
        } else
#ifdef UVRT_TRIP_STATS
            step7<TOP, FL>(L, p, stack_base, top_base, m_in, m_lf & kme, m_top, full, clk);
#else
            step7<TOP, FL>(L, p, stack_base, top_base, m_in, m_lf & kme, m_top, full);
#endif
    }
#ifdef UVRT_TRIP_STATS
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* st = (unsigned long long*)(p.error_flag + 2);
        const uint32_t v[26] = {st_trips, st_in, st_leaf, st_wait, st_idle, st_leaftrips, st_drain, st_slow, st_refills, st_top,
                                st_d9, st_d10, st_d11, st_d12, clk[0], clk[1], clk[2], clk[3],
                                (uint32_t)(__builtin_readcyclecounter() - t_begin), clk4,
                                st_s_in, st_s_leaf, st_s_both, st_g_exact, st_g_other, st_tris};
        for (int i = 0; i < 26; ++i) atomicAdd(&st[i + 1], (unsigned long long)v[i]);
        atomicAdd(&st[0], 1ull);
    }
#endif
    // the wave's sequence is exhausted and every lane is idle: deposit what is still pending
    if (RECORD && live && p.hits) {
        const uint32_t li = p.order ? p.order[slot] : slot;
        p.hits[li] = make_uint2(__float_as_uint(L.po.y), L.triID);
    }
    if (L.po.y != 1e30f) atomicAdd(&my_counts[(plane_off & PLANE_OFF6) + L.triID], 1);   // extend.cl:94-98
}

// Per-launch node-pair records recs[0, P): the lamp's x and z subtracted from the x / z bounds (the
// same single f32 subtraction IntersectAABB performs, extend.cl:31,35), one axis of one child per
// register pair, leaf references re-based to record indices (+P).
//   w0 = child0 {minx, maxx, minz, maxz}, w1 = child1 likewise,
//   w2 = {c0 miny, c0 maxy, c1 miny, c1 maxy} (raw), w3 = {ref0, ref1, 0, 0}
__global__ __launch_bounds__(256) void k_prepare_launch6(const PairRec* __restrict__ pairs,
                                                         float4* __restrict__ recs, float ox, float oz,
                                                         int32_t npairs, const uint32_t* __restrict__ perm)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= npairs) return;
    prepare_record6(pairs, recs, ox, oz, npairs, i, perm);
}

// Leaf-triangle records recs[P, P + T) (64 bytes each), once per scene
__global__ __launch_bounds__(256) void k_prepare_leaves6(const LeafTri* __restrict__ ltris,
                                                         float4* __restrict__ recs, int32_t npairs, int32_t T)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    const LeafTri t = ltris[i];
    float4* out = recs + ((size_t)npairs + i) * 4;
    out[0] = t.v0_id;
    out[1] = t.e1;
    out[2] = t.e2;
    out[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

void launch_prepare_launch6(const PairRec* pairs, void* recs, float ox, float oz, int32_t npairs, const uint32_t* perm,
                            hipStream_t s)
{
    if (npairs <= 0) return;
    hipLaunchKernelGGL(k_prepare_launch6, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, s, pairs, (float4*)recs,
                       ox, oz, npairs, perm);
}

void launch_prepare_leaves6(const LeafTri* ltris, void* recs, int32_t npairs, int32_t T, hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(k_prepare_leaves6, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, ltris,
                       (float4*)recs, npairs, T);
}

// code: bits 0-1 leaf period - 1 (0..3), bit 2 = WITHOUT the top-of-tree cache (codes other than 1: developer build only)
bool launch_extend6(const ExtendParams& p0, int code, int grid_per_cu, hipStream_t s)
{
    if (p0.n <= 0) return true;
    ExtendParams p = p0;
    const unsigned cus = p.num_cus > 0 ? (unsigned)p.num_cus : 256u;
    unsigned grid = cus * (unsigned)grid_per_cu;
    if (p.plane_batches == 0) {      // one launch: a single plane that holds all n rays
        p.plane_batches = (uint32_t)((p.n + 63) / 64);
        p.plane_n = (uint32_t)p.n;
        p.plane_stride = 0;
    }
    p.plane_inv = 1.0f / (float)p.plane_batches;
    p.refill_min = p.refill_min < 1 ? 1 : (p.refill_min > 64 ? 64 : p.refill_min);
    const unsigned need = (unsigned)((p.n + 255) / 256);
    if (need < grid) grid = need;
    const uint64_t waves = (uint64_t)grid * 4;
    p.chunk = (uint32_t)((((uint64_t)p.n + waves - 1) / waves + 63) / 64 * 64);
    if ((uint64_t)grid * 256 * (MAXS6 - PS6) > p.ovf_capacity) return false;
    // an inner root is translated through the renumbering by the kernel itself (perm lives on the device)
    p.root_ref6 = (p.scene.root_ref >= REF_LEAF_BIT && p.scene.root_ref != REF_DONE)
                      ? p.scene.root_ref + (uint32_t)p.npairs : p.scene.root_ref;
    if (p.perm) p.top_pairs = (uint32_t)p.npairs;   // the hot prefix: as many records as the cache holds
    if (p.npairs > 0 && !p.recs_prepared)
        hipLaunchKernelGGL(k_prepare_launch6, dim3((unsigned)((p.npairs + 255) / 256)), dim3(256), 0, s,
                           p.scene.pairs, (float4*)p.recs, p.ox, p.oz, p.npairs, p.perm);
#define UVRT_L6K(LP, REC, TOP, FL) hipLaunchKernelGGL((k_extend6<LP, REC, TOP, FL>), dim3(grid), dim3(256), 0, s, p)
#define UVRT_L6(LP, TOP)                                                                             \
    do {                                                                                             \
        if (p.flavour == 2) { if (p.hits) UVRT_L6K(LP, true, TOP, 2); else UVRT_L6K(LP, false, TOP, 2); }        \
        else if (p.flavour) { if (p.hits) UVRT_L6K(LP, true, TOP, 1); else UVRT_L6K(LP, false, TOP, 1); }       \
        else { if (p.hits) UVRT_L6K(LP, true, TOP, 0); else UVRT_L6K(LP, false, TOP, 0); }                     \
    } while (0)
#ifdef UVRT_DEV_VARIANTS     // developer build (`make` builds it as libuvrt_hip_dev.so): every leaf period, with / without the LDS cache
    switch (code & 7) {
        case 0: UVRT_L6(1, true); break;
        case 1: UVRT_L6(2, true); break;
        case 2: UVRT_L6(3, true); break;
        case 3: UVRT_L6(4, true); break;
        case 4: UVRT_L6(1, false); break;
        case 5: UVRT_L6(2, false); break;
        case 6: UVRT_L6(3, false); break;
        default: UVRT_L6(4, false); break;
    }
#else                        // the product: the default kernel only (leaf visits every second trip, LDS cache)
    if ((code & 7) != 1) return false;
    UVRT_L6(2, true);
#endif
#undef UVRT_L6K
#undef UVRT_L6
    return true;
}

}  // namespace uvrt
