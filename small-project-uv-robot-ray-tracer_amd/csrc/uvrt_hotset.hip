// uvrt_hotset.hip -- which node-pair records a lamp's photons visit most, found on the GPU.
//
// The traversal kernel (uvrt_extend6.hip) serves the first records of its numbering from LDS.  Which
// records are hot depends on the lamp: the 127 most visited ones take 62-70 % of all inner-node visits on
// the test room (the cache holds 175), the first 127 in breadth-first order 31-40 % (profiles/r02_record_layout_experiment.txt).
// For every new lamp position the context therefore
//   1. traces a sample of the launch's own photons (global ids [0, S)) with k_visit_stats -- a plain
//      one-ray-per-lane closest-hit traversal in fast arithmetic that only COUNTS inner-node visits, and
//   2. builds the renumbering with k_select_hot: the K most visited records first (ties by index), the rest
//      behind them in their old order.
// Only the ORDER of records in memory follows from these statistics; every result of the traversal proper is
// independent of it (a child reference is translated together with the records, uvrt_device.h
// prepare_record6), so nothing here has to be exact and nothing is synchronised with the host.
#include "uvrt_device.h"

namespace uvrt {

struct StatParams {
    const PairRec* pairs;
    const LeafTri* ltris;
    const uint32_t* leaf_count;
    uint32_t* hist;          // [npairs] visit counts, zero before the launch
    uint32_t root_ref;
    float lx, ly, lz, light_length;
    uint32_t seed_prev, seed_next;
    int32_t seed_mode;
    int32_t n;
};

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

__global__ __launch_bounds__(256) void k_visit_stats(StatParams p)
{
    const int gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= p.n || p.root_ref >= REF_LEAF_BIT) return;
    float r0;
    double sx, sy;
    const float4 ray = generate_ray(p.lx, p.ly, p.lz, p.light_length, gid, p.seed_prev, p.seed_next, p.seed_mode, r0, sx, sy);
    const float ox = p.lx, oy = ray.w, oz = p.lz;
    const float dx = ray.x, dy = ray.y, dz = ray.z;
    const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
    float dist = 1e30f;
    uint32_t stack[32];
    int sp = 0;
    uint32_t cur = p.root_ref;
    for (;;) {
        if (cur >= REF_LEAF_BIT) {
            const uint32_t first = cur & REF_FIRST_MASK;
            uint32_t count = (cur >> REF_COUNT_SHIFT) & 15u;
            if (count == 15u) count = p.leaf_count[first];
            for (uint32_t i = 0; i < count; ++i) {
                const LeafTri t = p.ltris[first + i];
                const float hx = dy * t.e2.z - dz * t.e2.y, hy = dz * t.e2.x - dx * t.e2.z, hz = dx * t.e2.y - dy * t.e2.x;
                const float a = t.e1.x * hx + t.e1.y * hy + t.e1.z * hz;
                if (fabsf(a) < 0.00001f) continue;
                const float f = 1.0f / a;
                const float qx0 = ox - t.v0_id.x, qy0 = oy - t.v0_id.y, qz0 = oz - t.v0_id.z;
                const float u = f * (qx0 * hx + qy0 * hy + qz0 * hz);
                if (u < 0 || u > 1) continue;
                const float qx = qy0 * t.e1.z - qz0 * t.e1.y, qy = qz0 * t.e1.x - qx0 * t.e1.z, qz = qx0 * t.e1.y - qy0 * t.e1.x;
                const float v = f * (dx * qx + dy * qy + dz * qz);
                if (v < 0 || u + v > 1) continue;
                const float tt = f * (t.e2.x * qx + t.e2.y * qy + t.e2.z * qz);
                if (tt > 0.0001f && tt < dist) dist = tt;
            }
            if (sp == 0) return;
            cur = stack[--sp];
            continue;
        }
        atomicAdd(&p.hist[cur], 1u);
        const PairRec pr = p.pairs[cur];
        float d0, d1;
        const bool h0 = box_approx(pr.c0min_ref0.x, pr.c0min_ref0.y, pr.c0min_ref0.z, pr.c0max_ref1.x, pr.c0max_ref1.y,
                                   pr.c0max_ref1.z, ox, oy, oz, ix, iy, iz, dist, d0);
        const bool h1 = box_approx(pr.c1min.x, pr.c1min.y, pr.c1min.z, pr.c1max.x, pr.c1max.y, pr.c1max.z, ox, oy, oz,
                                   ix, iy, iz, dist, d1);
        const uint32_t r0r = __float_as_uint(pr.c0min_ref0.w), r1r = __float_as_uint(pr.c0max_ref1.w);
        if (h0 && h1) {
            const bool sw = d0 > d1;
            if (sp < 32) stack[sp++] = sw ? r0r : r1r;
            cur = sw ? r1r : r0r;
        } else if (h0 || h1) {
            cur = h0 ? r0r : r1r;
        } else {
            if (sp == 0) return;
            cur = stack[--sp];
        }
    }
}

// perm[i] = new index of record i: the `keep` most visited records first (in index order among themselves,
// ties at the threshold broken by index), all others behind them in index order.  One workgroup.
__global__ __launch_bounds__(1024) void k_select_hot(uint32_t* __restrict__ hist, uint32_t* __restrict__ perm, int32_t n,
                                                     int32_t keep)
{
    __shared__ uint32_t s_cnt[1024];
    __shared__ uint32_t s_lo, s_hi;
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int b = tid * per, e = min(n, b + per);
    auto block_sum = [&](uint32_t v) -> uint32_t {      // inclusive scan in s_cnt, returns the total
        s_cnt[tid] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t t = tid >= off ? s_cnt[tid - off] : 0u;
            __syncthreads();
            s_cnt[tid] += t;
            __syncthreads();
        }
        return s_cnt[1023];
    };
    // the smallest threshold t with #(count > t) <= keep, by bisection over the count values
    if (tid == 0) { s_lo = 0u; s_hi = 0xFFFFFFFFu; }
    __syncthreads();
    for (int it = 0; it < 32; ++it) {
        const uint32_t lo = s_lo, hi = s_hi;
        if (lo >= hi) break;
        const uint32_t mid = lo + (hi - lo) / 2;
        uint32_t c = 0;
        for (int i = b; i < e; ++i) c += hist[i] > mid;
        const uint32_t above = block_sum(c);
        if (tid == 0) { if ((int32_t)min(above, 0x7FFFFFFFu) <= keep) s_hi = mid; else s_lo = mid + 1; }
        __syncthreads();
    }
    const uint32_t thr = s_lo;
    // records above the threshold are hot; those equal to it fill what is left, lowest index first
    uint32_t c_above = 0, c_equal = 0;
    for (int i = b; i < e; ++i) { c_above += hist[i] > thr; c_equal += hist[i] == thr; }
    const uint32_t tot_above = block_sum(c_above);
    const uint32_t pre_above = s_cnt[tid] - c_above;
    __syncthreads();
    (void)block_sum(c_equal);
    const uint32_t pre_equal = s_cnt[tid] - c_equal;
    __syncthreads();
    const uint32_t room = (uint32_t)keep > tot_above ? (uint32_t)keep - tot_above : 0u;   // ties admitted
    // hot rank = (#hot with a smaller index); cold rank likewise: two more prefix sums over the final flags
    uint32_t c_hot = 0;
    {
        uint32_t eq = pre_equal;
        for (int i = b; i < e; ++i) {
            const uint32_t h = hist[i];
            c_hot += (h > thr) || (h == thr && eq < room);
            eq += h == thr;
        }
    }
    const uint32_t tot_hot = block_sum(c_hot);
    uint32_t hot_before = s_cnt[tid] - c_hot;
    __syncthreads();
    {
        uint32_t eq = pre_equal;
        uint32_t cold_before = (uint32_t)b - hot_before;
        for (int i = b; i < e; ++i) {
            const uint32_t h = hist[i];
            const bool hot = (h > thr) || (h == thr && eq < room);
            eq += h == thr;
            perm[i] = hot ? hot_before++ : tot_hot + cold_before++;
        }
    }
    (void)pre_above;
    __syncthreads();
    for (int i = b; i < e; ++i) hist[i] = 0u;          // ready for the next lamp
}

void launch_visit_stats(const SceneDev& scene, uint32_t* hist, const float lamp[3], float light_length, uint32_t seed_prev,
                        uint32_t seed_next, int32_t seed_mode, int32_t n, hipStream_t s)
{
    if (n <= 0) return;
    StatParams p;
    p.pairs = scene.pairs;
    p.ltris = scene.ltris;
    p.leaf_count = scene.leaf_count;
    p.hist = hist;
    p.root_ref = scene.root_ref;
    p.lx = lamp[0]; p.ly = lamp[1]; p.lz = lamp[2];
    p.light_length = light_length;
    p.seed_prev = seed_prev;
    p.seed_next = seed_next;
    p.seed_mode = seed_mode;
    p.n = n;
    hipLaunchKernelGGL(k_visit_stats, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p);
}

void launch_select_hot(uint32_t* hist, uint32_t* perm, int32_t npairs, int32_t keep, hipStream_t s)
{
    if (npairs <= 0) return;
    hipLaunchKernelGGL(k_select_hot, dim3(1), dim3(1024), 0, s, hist, perm, npairs, keep);
}

}  // namespace uvrt
