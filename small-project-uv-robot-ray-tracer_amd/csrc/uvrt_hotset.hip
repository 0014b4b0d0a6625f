// uvrt_hotset.hip -- which node-pair records a lamp's photons visit most, found on the GPU.
//
// The traversal kernel (uvrt_extend6.hip) serves the first records of its numbering from LDS.  Which
// records are hot depends on the lamp: the 127 most visited ones take 62-70 % of all inner-node visits on
// the test room (the cache holds 175), the first 127 in breadth-first order 31-40 % (profiles/r02/r02_record_layout_experiment.txt).
// For every new lamp position the context therefore enqueues three small kernels on the launch's stream (the new lamps of
// one uvrt_trace_batch call share ONE launch of each: blockIdx.y = the lamp):
//   1. k_visit_stats -- traces a sample of the launch's own photons (global ids [0, S)): a plain
//      one-ray-per-lane closest-hit traversal in fast arithmetic that only COUNTS inner-node visits.  The
//      counters of the top of the tree (breadth-first indices < HS_LDS_BINS, where every ray of a workgroup
//      meets) are privatised in LDS and flushed once per workgroup, visits to deeper records are queued in LDS and
//      counted in bursts; the traversal stack lives in LDS too.
//      (Round 2 counted every visit with a global atomic: 32 768 same-address atomics on the root's counter
//      alone made the kernel 2.2 ms long -- longer than a whole 8-wave step.)
//   2. k_select_hot -- ONE workgroup, work independent of the scene size: a ray visits a node only after
//      its parent, so count(child) <= count(parent) and the most visited records form a subtree that
//      contains the root.  The kernel grows that subtree level by level from the root (children whose count
//      reaches a floor join the candidates, at most HS_CAND of them), finds the count of the K-th largest by a
//      radix select over the candidates, breaks ties at that count by index and writes the hot records' indices
//      in ascending order.
//   3. k_write_perm -- the renumbering for all records (one thread per record, binary search in the hot
//      list): the hot records first (in index order among themselves), all others behind them in index order;
//      the visit counters are zeroed again for the next lamp.
// Only the ORDER of records in memory follows from these statistics; every result of the traversal proper is
// independent of it (a child reference is translated together with the records, uvrt_device.h
// prepare_record6), so nothing here has to be exact and nothing is synchronised with the host.
#include "uvrt_device.h"

namespace uvrt {

constexpr int HS_LDS_BINS = 8192;     // visit counters kept in LDS by k_visit_stats (the first 13 tree levels)
constexpr int HS_QUEUE = 16;          // visits to deeper records a lane collects in LDS before the wave flushes them
constexpr int HS_CAND = 4096;         // candidate records of k_select_hot
constexpr int HS_EQ = 1024;           // candidates AT the threshold that take part in the tie-break by index
constexpr int HS_MAX_STEPS = 128;     // traversal steps of a sample ray that are counted at most

// Several lamps' set-ups in one launch of each kernel (uvrt_device.h HotSetupParams): blockIdx.y = the lamp.
typedef HotSetupParams StatParams;

__device__ __forceinline__ bool box_approx(float mnx, float mny, float mnz, float mxx, float mxy, float mxz, float ox,
                                           float oy, float oz, float ix, float iy, float iz, float dist, float& tmin)
{
    const float tx1 = (mnx - ox) * ix, tx2 = (mxx - ox) * ix;
    const float ty1 = (mny - oy) * iy, ty2 = (mxy - oy) * iy;
    const float tz1 = (mnz - oz) * iz, tz2 = (mxz - oz) * iz;
    tmin = fmaxf(fmaxf(fminf(tx1, tx2), fminf(ty1, ty2)), fminf(tz1, tz2));
    const float tmax = fminf(fminf(fmaxf(tx1, tx2), fmaxf(ty1, ty2)), fmaxf(tz1, tz2));
    return tmax >= tmin && tmin < dist && tmax > 0;
}

// (Workgroups of 128 or 64 threads, spread over twice / four times as many CUs, are no faster: 98 / 106 us against 97 us --
// the kernel lasts as long as the dependent fetches of its longest rays, profiles/r03/r03_hot_setup_sweep.txt.)
constexpr int HS_THREADS = 256;
__global__ __launch_bounds__(HS_THREADS) void k_visit_stats(StatParams p)
{
    constexpr int NT = HS_THREADS;
    __shared__ uint32_t s_hist[HS_LDS_BINS];      // 32 KB
    __shared__ uint32_t s_stack[32][NT];          // 32 KB: entry e of thread t at [e][t] (conflict-free)
    __shared__ uint32_t s_queue[HS_QUEUE][NT];    // 16 KB: deep records visited by thread t, not yet counted
    const int tid = threadIdx.x;
    for (int i = tid; i < HS_LDS_BINS; i += NT) s_hist[i] = 0u;
    __syncthreads();
    const int gid = blockIdx.x * NT + tid;
    const int grp = blockIdx.y;                                       // the lamp of this workgroup
    uint32_t* const hist = p.hist + (size_t)grp * (size_t)p.npairs;
    uint32_t cur = (gid < p.n && p.root_ref < REF_LEAF_BIT) ? p.root_ref : REF_DONE;
    float r0;
    double sx, sy;
    const float4 ray = generate_ray(p.lx[grp], p.ly[grp], p.lz[grp], p.light_length, gid, p.seed_prev[grp], p.seed_next[grp], p.seed_mode,
                                    r0, sx, sy);
    const float ox = p.lx[grp], oy = ray.w, oz = p.lz[grp];
    const float dx = ray.x, dy = ray.y, dz = ray.z;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    float dist = 1e30f;
    int sp = 0;
    int qn = 0;
    // A visit to a record beyond the LDS counters is queued in LDS and counted later, HS_QUEUE visits per lane in one
    // burst of global atomics: memory operations complete in issue order (vmcnt), so an atomic issued in every trip
    // would put its round trip to the memory side (600-3000 cycles) in front of the next trip's record fetch.
    auto flush_queue = [&]() {
        for (int e = 0; e < HS_QUEUE; ++e)
            if (e < qn) atomicAdd(&hist[s_queue[e][tid]], 1u);
        qn = 0;
    };
    // One step per trip, ONE memory round trip per step: the 64 bytes at the lane's record -- a node-pair record or a
    // leaf triangle (48 bytes; the 16 behind it are the next triangle's or the buffer's padding).  Deep records are
    // cold by definition, so every trip of a wave waits for a miss to HBM: the kernel lasts (steps of the slowest
    // ray) x (miss latency + the step's ~150 instructions at one wave per SIMD).  So a wave does not wait for its
    // stragglers: it stops when at most p.tail_lanes of its rays are still under way (16 of 64: the slowest quarter of the
    // rays takes half as many steps again as the rest; what those would still visit does not change which records are hot --
    // coverage of the true best 175 over the route's 12 lamps 0.9989-0.9999 with 16, 0.9994-0.9999 with 6, and 10 % less
    // time; profiles/r03/r03_hot_setup_sweep.txt).
    for (int it = 0; it < HS_MAX_STEPS && cur != REF_DONE; ++it) {
        if (it >= 32 && __popcll(__builtin_amdgcn_ballot_w64(true)) <= p.tail_lanes) break;     // (the lanes still in the loop)
        const bool leaf = cur >= REF_LEAF_BIT;
        const uint32_t first = cur & REF_FIRST_MASK;
        const float4* src = leaf ? (const float4*)(p.ltris + first) : (const float4*)(p.pairs + cur);
        const float4 w0 = src[0], w1 = src[1], w2 = src[2], w3 = src[3];
        if (leaf) {
            uint32_t count = (cur >> REF_COUNT_SHIFT) & 15u;
            if (count == 15u) count = p.leaf_count[first];
            float4 v0 = w0, e1 = w1, e2 = w2;
            for (uint32_t i = 0;;) {
                const float hx = dy * e2.z - dz * e2.y, hy = dz * e2.x - dx * e2.z, hz = dx * e2.y - dy * e2.x;
                const float a = e1.x * hx + e1.y * hy + e1.z * hz;
                if (fabsf(a) >= 0.00001f) {
                    const float f = 1.0f / a;
                    const float qx0 = ox - v0.x, qy0 = oy - v0.y, qz0 = oz - v0.z;
                    const float u = f * (qx0 * hx + qy0 * hy + qz0 * hz);
                    const float qx = qy0 * e1.z - qz0 * e1.y, qy = qz0 * e1.x - qx0 * e1.z, qz = qx0 * e1.y - qy0 * e1.x;
                    const float v = f * (dx * qx + dy * qy + dz * qz);
                    const float tt = f * (e2.x * qx + e2.y * qy + e2.z * qz);
                    if (u >= 0 && u <= 1 && v >= 0 && u + v <= 1 && tt > 0.0001f && tt < dist) dist = tt;
                }
                if (++i >= count) break;
                const LeafTri t = p.ltris[first + i];
                v0 = t.v0_id; e1 = t.e1; e2 = t.e2;
            }
            cur = sp > 0 ? s_stack[--sp][tid] : REF_DONE;
        } else {
            if (cur < (uint32_t)HS_LDS_BINS) atomicAdd(&s_hist[cur], 1u);
            else s_queue[qn++][tid] = cur;
            // PairRec: w0 = child 0 min + ref0, w1 = child 0 max + ref1, w2 = child 1 min, w3 = child 1 max
            float d0, d1;
            const bool h0 = box_approx(w0.x, w0.y, w0.z, w1.x, w1.y, w1.z, ox, oy, oz, ix, iy, iz, dist, d0);
            const bool h1 = box_approx(w2.x, w2.y, w2.z, w3.x, w3.y, w3.z, ox, oy, oz, ix, iy, iz, dist, d1);
            const uint32_t r0r = __float_as_uint(w0.w), r1r = __float_as_uint(w1.w);
            if (h0 && h1) {
                const bool sw = d0 > d1;
                if (sp < 32) s_stack[sp++][tid] = sw ? r0r : r1r;
                cur = sw ? r1r : r0r;
            } else if (h0 || h1) {
                cur = h0 ? r0r : r1r;
            } else {
                cur = sp > 0 ? s_stack[--sp][tid] : REF_DONE;
            }
        }
        if (__builtin_amdgcn_ballot_w64(qn >= HS_QUEUE) != 0) flush_queue();      // wave-uniform
    }
    flush_queue();
    __syncthreads();
    for (int i = tid; i < HS_LDS_BINS; i += NT) {
        const uint32_t v = s_hist[i];
        if (v) atomicAdd(&hist[i], v);            // only bins below npairs are ever counted
    }
}

// hot[0] = H (number of hot records, <= keep), hot[1 .. H] = their indices in ascending order.  One workgroup.
constexpr int HS_SEL_THREADS = 256;
__global__ __launch_bounds__(HS_SEL_THREADS) void k_select_hot(const PairRec* __restrict__ pairs, const uint32_t* __restrict__ hist_all,
                                                             uint32_t* __restrict__ hot_all, int32_t npairs, int32_t keep, int32_t direct_bins)
{
    const uint32_t* __restrict__ hist = hist_all + (size_t)blockIdx.x * (size_t)npairs;      // one workgroup per lamp
    uint32_t* __restrict__ hot = hot_all + (size_t)blockIdx.x * (TOP6_MAX + 1);
    __shared__ uint32_t c_idx[HS_CAND], c_cnt[HS_CAND];
    __shared__ uint32_t e_idx[HS_EQ];                 // candidates at the threshold
    __shared__ uint32_t h_idx[TOP6_MAX + 1];          // the hot records, unordered
    __shared__ uint32_t s_bins[256], s_wsum[4];
    __shared__ uint32_t s_n, s_begin, s_end, s_ne, s_nh, s_digit, s_above;
    const int tid = threadIdx.x;
    constexpr uint32_t NT = HS_SEL_THREADS;
    const uint32_t root_cnt = hist[0];                // every sampled ray visits the root (pair record 0)
    // The counters of the first direct_bins records, HS_LDS_BINS / NT per thread in registers (one coalesced read): most of
    // the hot records lie among them, and there no tree walk is needed -- any record whose count reaches the floor is a
    // candidate (the counts themselves make the selection a subtree).  Only the part of the subtree that reaches beyond
    // them is walked, from the candidates whose children lie there (every other lamp of the test room has such a part).
    const uint32_t nd = (uint32_t)npairs < (uint32_t)direct_bins ? (uint32_t)npairs : (uint32_t)direct_bins;      // direct_bins <= HS_LDS_BINS
    uint32_t mine[HS_LDS_BINS / HS_SEL_THREADS];
#pragma unroll
    for (int k = 0; k < HS_LDS_BINS / HS_SEL_THREADS; ++k) {
        const uint32_t j = (uint32_t)k * NT + (uint32_t)tid;
        mine[k] = j < nd ? hist[j] : 0u;
    }
    // the floor a record's count must reach to become a candidate; lowered if it leaves fewer than `keep`
    uint32_t floor_cnt = root_cnt >> 6;
    if (floor_cnt < 1u) floor_cnt = 1u;
    uint32_t M;
    for (;;) {
        __syncthreads();
        if (tid == 0) { c_idx[0] = 0u; c_cnt[0] = root_cnt; s_n = nd ? 0u : 1u; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < HS_LDS_BINS / HS_SEL_THREADS; ++k) {
            if (mine[k] < floor_cnt) continue;            // (mine[] is 0 beyond nd, floor_cnt >= 1)
            const uint32_t slot = atomicAdd(&s_n, 1u);
            if (slot < (uint32_t)HS_CAND) { c_idx[slot] = (uint32_t)k * NT + (uint32_t)tid; c_cnt[slot] = mine[k]; }
        }
        __syncthreads();
        if (tid == 0) { s_begin = 0u; s_end = s_n < (uint32_t)HS_CAND ? s_n : (uint32_t)HS_CAND; s_n = s_end; }
        __syncthreads();
        // the part of the hot subtree beyond the first nd records, one tree level per round (with nd = 0: the whole walk
        // from the root): children at indices >= nd whose count reaches the floor
        for (;;) {
            const uint32_t b = s_begin, e = s_end;
            if (b >= e || (uint32_t)npairs <= nd) break;
            for (uint32_t j = b + tid; j < e; j += NT) {
                const PairRec* pr = pairs + c_idx[j];
                const uint32_t r[2] = {__float_as_uint(pr->c0min_ref0.w), __float_as_uint(pr->c0max_ref1.w)};
                for (int k = 0; k < 2; ++k) {
                    if (r[k] >= REF_LEAF_BIT || r[k] < nd) continue;
                    const uint32_t c = hist[r[k]];
                    if (c < floor_cnt) continue;
                    const uint32_t slot = atomicAdd(&s_n, 1u);
                    if (slot < (uint32_t)HS_CAND) { c_idx[slot] = r[k]; c_cnt[slot] = c; }
                }
            }
            __syncthreads();
            if (tid == 0) { s_begin = e; s_end = s_n < (uint32_t)HS_CAND ? s_n : (uint32_t)HS_CAND; s_n = s_end; }
            __syncthreads();
        }
        M = s_end;
        if (M >= (uint32_t)keep || floor_cnt == 1u) break;
        floor_cnt = floor_cnt > 8u ? floor_cnt >> 3 : 1u;
    }
    // The count of the keep-th largest candidate (0 when there are fewer), digit by digit: per 8-bit digit a histogram
    // of the candidates that match the digits fixed so far, its suffix sums (bin 255 down) by a scan over the 256
    // threads, and the bin in which `keep` candidates are reached.  Then #(count > thr) < keep.
    uint32_t prefix = 0u, above = 0u;                 // digits fixed so far; candidates known to lie above the threshold
    const int top_shift = root_cnt < (1u << 8) ? 0 : root_cnt < (1u << 16) ? 8 : root_cnt < (1u << 24) ? 16 : 24;
    const uint32_t lane = (uint32_t)tid & 63u, wv = (uint32_t)tid >> 6;
    for (int shift = top_shift; shift >= 0; shift -= 8) {
        s_bins[tid] = 0u;                             // NT == 256 bins
        __syncthreads();
        const uint32_t hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (uint32_t j = tid; j < M; j += NT) {
            const uint32_t c = c_cnt[j];
            if ((c & hi_mask) == prefix) atomicAdd(&s_bins[(c >> shift) & 255u], 1u);
        }
        __syncthreads();
        const uint32_t v = s_bins[255 - tid];         // thread t looks at digit 255 - t
        uint32_t x = v;                               // inclusive scan: candidates with a digit >= 255 - t
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off, 64);
            if ((int)lane >= off) x += y;
        }
        if (lane == 63u) s_wsum[wv] = x;
        __syncthreads();
        for (uint32_t w2 = 0; w2 < wv; ++w2) x += s_wsum[w2];
        const uint32_t upto = above + x;              // candidates above the threshold's range or with a digit >= 255 - t
        if ((upto >= (uint32_t)keep && upto - v < (uint32_t)keep) || (tid == 255 && upto < (uint32_t)keep)) {
            s_digit = 255u - (uint32_t)tid;           // exactly one thread: the digit of the keep-th largest (0 if too few)
            s_above = upto - v;
        }
        __syncthreads();
        prefix |= s_digit << shift;
        above = s_above;
    }
    const uint32_t thr = prefix;                      // count > thr: hot for sure (`above` of them, < keep); == thr: ties
    if (tid == 0) { s_ne = 0u; s_nh = 0u; }
    __syncthreads();
    for (uint32_t j = tid; j < M; j += NT) {
        const uint32_t c = c_cnt[j];
        if (c > thr) h_idx[atomicAdd(&s_nh, 1u)] = c_idx[j];
        else if (c == thr) { const uint32_t q = atomicAdd(&s_ne, 1u); if (q < (uint32_t)HS_EQ) e_idx[q] = c_idx[j]; }
    }
    __syncthreads();
    const uint32_t n_above = s_nh;
    const uint32_t n_eq = s_ne < (uint32_t)HS_EQ ? s_ne : (uint32_t)HS_EQ;
    const uint32_t room = (uint32_t)keep - n_above;                     // ties admitted, lowest index first
    for (uint32_t j = tid; j < n_eq; j += NT) {
        const uint32_t mine = e_idx[j];
        uint32_t before = 0u;
        for (uint32_t k = 0; k < n_eq; ++k) before += e_idx[k] < mine;
        if (before < room) h_idx[n_above + before] = mine;
    }
    __syncthreads();
    const uint32_t H = n_above + (n_eq < room ? n_eq : room);
    for (uint32_t j = tid; j < H; j += NT) {
        const uint32_t mine_idx = h_idx[j];
        uint32_t before = 0u;
        for (uint32_t k = 0; k < H; ++k) before += h_idx[k] < mine_idx;
        hot[1u + before] = mine_idx;
    }
    if (tid == 0) hot[0] = H;
}

// perm[i] = new index of record i: the hot records first (in index order among themselves), all others behind
// them in index order; the visit counters zeroed for the next lamp.
__global__ __launch_bounds__(256) void k_write_perm(HotSetupParams p)
{
    const uint32_t* __restrict__ hot = p.hot_list + (size_t)blockIdx.y * (TOP6_MAX + 1);
    uint32_t* __restrict__ perm = p.perm[blockIdx.y];
    uint32_t* __restrict__ hist = p.hist + (size_t)blockIdx.y * (size_t)p.npairs;
    const int32_t n = p.npairs;
    __shared__ uint32_t s_hot[TOP6_MAX + 1];
    const uint32_t H = hot[0] <= TOP6_MAX ? hot[0] : TOP6_MAX;
    for (uint32_t j = threadIdx.x; j < H; j += 256u) s_hot[j] = hot[1u + j];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t lo = 0u, hi = H;                          // hot records with an index below i
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (s_hot[mid] < (uint32_t)i) lo = mid + 1u; else hi = mid;
    }
    const bool is_hot = lo < H && s_hot[lo] == (uint32_t)i;
    perm[i] = is_hot ? lo : H + ((uint32_t)i - lo);
    hist[i] = 0u;
}

void launch_hot_setup(const HotSetupParams& p, hipStream_t s)
{
    if (p.count <= 0 || p.npairs <= 0 || p.n <= 0) return;
    hipLaunchKernelGGL(k_visit_stats, dim3((unsigned)((p.n + HS_THREADS - 1) / HS_THREADS), (unsigned)p.count), dim3(HS_THREADS), 0, s, p);
    hipLaunchKernelGGL(k_select_hot, dim3((unsigned)p.count), dim3(HS_SEL_THREADS), 0, s, p.pairs, (const uint32_t*)p.hist, p.hot_list,
                       p.npairs, p.keep < (int32_t)TOP6_MAX ? p.keep : (int32_t)TOP6_MAX,
                       p.direct_bins < 0 ? 0 : p.direct_bins < HS_LDS_BINS ? p.direct_bins : HS_LDS_BINS);
    hipLaunchKernelGGL(k_write_perm, dim3((unsigned)((p.npairs + 255) / 256), (unsigned)p.count), dim3(256), 0, s, p);
}

}  // namespace uvrt
