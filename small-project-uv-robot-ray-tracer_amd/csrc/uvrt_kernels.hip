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
