// uvrt_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the UV-dose hot path.
//
// Compile with -ffp-contract=off and correctly rounded f32 divide/sqrt (csrc/Makefile): every
// result must equal the reference kernels' strict-IEEE arithmetic bit for bit, so each
// operator below is one rounding in the order the reference source writes it.
//
//   k_generate        cl/generate.cl:8-40       + optional coherence key / rank, extend's per-launch records
//   k_scan_bins, k_scatter                      optional counting sort of rays by coherence key
//   (k_extend6        cl/extend.cl:6-99         BVH traversal + photon deposit, THE hot loop: uvrt_extend6.hip)
//   k_accumulate      cl/accumulate.cl:4-14
//   k_reset           cl/reset.cl:4-26
//   k_compute_dosage  cl/shade.cl:23-41
//   k_dosage_to_color cl/shade.cl:4-21,43-71
#include "uvrt_device.h"

namespace uvrt {

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
    float r0;
    double x, y;
    const float4 ray = generate_ray(p.lx, p.ly, p.lz, p.light_length, gid, p.seed_prev, p.seed_next, p.seed_mode, r0, x, y);
    const float diry = ray.y;
    p.rays[i] = ray;                                                      // :31-37

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

// The rays of several launches side by side (uvrt_trace_batch): blockIdx.y = physical plane, every plane
// generates the same global-id range under its own lamp and place in the SEED chain.
__global__ __launch_bounds__(256) void k_generate_batch(GenBatchParams p)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const int k = blockIdx.y;
    float r0;
    double x, y;
    p.rays[(int64_t)k * p.n_pad + i] = generate_ray(p.lx[k], p.ly[k], p.lz[k], p.light_length, p.first_gid + i,
                                                    p.seed_prev[k], p.seed_next[k], p.seed_mode, r0, x, y);
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
                                                 uint32_t* __restrict__ order, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint2 kr = keyrank[i];
    const uint32_t pos = bin_start[kr.x] + kr.y;
    const float4 r = rays[i];
    sorted[pos] = r;
    order[pos] = (uint32_t)i;
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

// Batched tracing: per plane the (exact, integer) sum of its deposit replicas -- the int32 payload of the one
// all-reduce of a ray-range-sharded computation -- and the replicas zeroed for the next batch.
__global__ __launch_bounds__(256) void k_fold_planes(int32_t* __restrict__ planes, int32_t* __restrict__ folded,
                                                     int32_t replicas, int32_t T)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int i = g >> 2, part = g & 3;
    int32_t* pl = planes + (int64_t)blockIdx.y * replicas * T;
    int32_t total = 0;
    if (i < T) {
        for (int r = part; r < replicas; r += 4) {
            total += pl[(int64_t)r * T + i];
            pl[(int64_t)r * T + i] = 0;
        }
    }
    total += __builtin_amdgcn_mov_dpp(total, 0xb1, 0xf, 0xf, true);   // quad_perm [1,0,3,2]
    total += __builtin_amdgcn_mov_dpp(total, 0x4e, 0xf, 0xf, true);   // quad_perm [2,3,0,1]
    if (i < T && part == 0) folded[(int64_t)blockIdx.y * T + i] = total;
}

// Replay of a batch in LOGICAL launch order, one quad of lanes per triangle: for every launch the
// reference's accumulate (cl/accumulate.cl:4-14: same f64 additions in the same order, max over the
// per-launch totals) and, where the host loop has a Shade behind it (raytracer.cpp:93-120), computeDosage +
// dosageToColor with that Shade's arguments.  The planes are left zeroed.
__global__ __launch_bounds__(256) void k_replay_batch(ReplayParams p)
{
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int i = g >> 2, part = g & 3;
    const bool in = i < p.T;
    double pm = 0, mm = 0;
    if (in && part == 0) { pm = p.photon_map[i]; mm = p.max_map[i]; }
    for (int k = 0; k < p.count; ++k) {
        const ReplayOp op = p.ops[k];
        int32_t total = 0;
        if (p.is_folded) {
            if (in && part == 0) {
                int32_t* f = p.folded + (int64_t)op.plane * p.T + i;
                total = *f;
                *f = 0;
            }
        } else {
            if (in) {
                int32_t* pl = p.planes + (int64_t)op.plane * p.plane_stride;
                for (int r = part; r < p.replicas; r += 4) {
                    total += pl[(int64_t)r * p.T + i];
                    pl[(int64_t)r * p.T + i] = 0;
                }
            }
            total += __builtin_amdgcn_mov_dpp(total, 0xb1, 0xf, 0xf, true);
            total += __builtin_amdgcn_mov_dpp(total, 0x4e, 0xf, 0xf, true);
        }
        if (!in || part != 0) continue;
        const double c = (double)total;                                   // accumulate.cl:9-13
        pm = pm + c * (double)op.duration;
        mm = mm < c ? c : mm;
        if (op.shade) {                                                   // shade.cl:23-41, 43-71
            const double num = (double)op.scaled_power * (op.which_map ? mm : pm);
            const float den = p.area[i] * (float)op.photons_per_light;
            const float dose = (float)(num / (double)den);
            p.dosage[i] = dose;
            const float maxValue = op.min_value * 2;
            const float norm = dose / maxValue;
            float r, gg, b;
            if (op.threshold_view && norm < 0.5f) { r = 0.0f; gg = 0.0f; b = norm * 2.0f; }
            else heatmap(norm, r, gg, b);
            float* col = p.color + (int64_t)i * 9;
            col[0] = r; col[1] = gg; col[2] = b;
            col[3] = r; col[4] = gg; col[5] = b;
            col[6] = r; col[7] = gg; col[8] = b;
        }
    }
    if (in && part == 0) { p.photon_map[i] = pm; p.max_map[i] = mm; }
}

__global__ __launch_bounds__(256) void k_add_counts(int32_t* __restrict__ dst, const int32_t* __restrict__ src, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] += src[i];
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

// computeDosage + dosageToColor of triangle i from its map value (shade.cl:23-71); the dose still goes through its f32 store
__device__ __forceinline__ void shade_one(int i, double map_value, float* __restrict__ dosage, const float* __restrict__ area,
                                          float* __restrict__ color, int32_t photons_per_light, float scaled_power,
                                          float min_value, int32_t threshold_view)
{
    const double num = (double)scaled_power * map_value;                // shade.cl:39
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

// computeDosage + dosageToColor in one pass (RayTracer::Shade always runs them back to back,
// raytracer.cpp:93-120); same operations
__global__ __launch_bounds__(256) void k_shade(const double* __restrict__ map, float* __restrict__ dosage,
                                               const float* __restrict__ area, float* __restrict__ color,
                                               int32_t photons_per_light, float scaled_power,
                                               float min_value, int32_t threshold_view, int32_t T)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T) return;
    shade_one(i, map[i], dosage, area, color, photons_per_light, scaled_power, min_value, threshold_view);
}

// accumulate.cl:4-14 of the launch just traced and the Shade that the host loop runs right after it (myapp.cpp:159-160)
// in ONE launch: k_accumulate's fold of the deposit replicas and its map update, then shade_one on the value just written.
// which_map: 0 photonMap, 1 maxPhotonMap.
__global__ __launch_bounds__(256) void k_accumulate_shade(double* __restrict__ photon_map, double* __restrict__ max_map,
                                                          int32_t* __restrict__ counts, int32_t replicas, int64_t stride,
                                                          float time_step, float* __restrict__ dosage,
                                                          const float* __restrict__ area, float* __restrict__ color,
                                                          int32_t which_map, int32_t photons_per_light, float scaled_power,
                                                          float min_value, int32_t threshold_view, int32_t T)
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
    const double sum = photon_map[i] + c * (double)time_step;          // accumulate.cl:9
    photon_map[i] = sum;
    const double m = max_map[i];
    const double mx = m < c ? c : m;                                    // accumulate.cl:11
    max_map[i] = mx;
    shade_one(i, which_map ? mx : sum, dosage, area, color, photons_per_light, scaled_power, min_value, threshold_view);
}

// Measurement hook (uvrt_clock_probe_start): one wave reads the shader-clock counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime) `ticks` of the latter apart, sleeping in between; out = {shader ticks, 100 MHz ticks}.  Launched on a stream of
// its own beside whatever the device is doing, it tells what the shader clock IS under that load (the issue-rate peaks of the
// bench line's roofline scale with it; it moves between 2.0 and 2.4 GHz with the power the load draws).
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

void launch_clock_probe(unsigned long long* out, unsigned long long ticks, hipStream_t s)
{
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, out, ticks);
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

void launch_generate_batch(const GenBatchParams& p, hipStream_t s)
{
    if (p.n <= 0 || p.count <= 0) return;
    hipLaunchKernelGGL(k_generate_batch, dim3(blocks_for(p.n, 256), (unsigned)p.count), dim3(256), 0, s, p);
}

void launch_fold_planes(int32_t* planes, int32_t* folded, int32_t nplanes, int32_t replicas, int32_t T, hipStream_t s)
{
    if (nplanes <= 0 || T <= 0) return;
    hipLaunchKernelGGL(k_fold_planes, dim3(blocks_for((int64_t)T * 4, 256), (unsigned)nplanes), dim3(256), 0, s, planes,
                       folded, replicas, T);
}

void launch_replay_batch(const ReplayParams& p, hipStream_t s)
{
    if (p.count <= 0 || p.T <= 0) return;
    hipLaunchKernelGGL(k_replay_batch, dim3(blocks_for((int64_t)p.T * 4, 256)), dim3(256), 0, s, p);
}

void launch_add_counts(int32_t* dst, const int32_t* src, int64_t n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_add_counts, dim3(blocks_for(n, 256)), dim3(256), 0, s, dst, src, n);
}

void launch_scan_bins(uint32_t* hist, uint32_t* bin_start, int32_t nbins, hipStream_t s)
{
    hipLaunchKernelGGL(k_scan_bins, dim3(1), dim3(1024), 0, s, hist, bin_start, nbins);
}

void launch_scatter(const float4* rays, const uint2* keyrank, const uint32_t* bin_start,
                    float4* sorted, uint32_t* order, int64_t n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n, 256)), dim3(256), 0, s, rays, keyrank,
                       bin_start, sorted, order, n);
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

void launch_accumulate_shade(double* photon_map, double* max_map, int32_t* counts, int32_t replicas, int64_t stride,
                             float time_step, float* dosage, const float* area, float* color, int32_t which_map,
                             int32_t photons_per_light, float scaled_power, float min_value, int32_t threshold_view,
                             int32_t T, hipStream_t s)
{
    if (T <= 0) return;
    hipLaunchKernelGGL(k_accumulate_shade, dim3(blocks_for((int64_t)T * 4, 256)), dim3(256), 0, s, photon_map, max_map, counts,
                       replicas, stride, time_step, dosage, area, color, which_map, photons_per_light, scaled_power, min_value,
                       threshold_view, T);
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
