// fetch_calib.hip -- what FETCH_SIZE / TCC_EA0_RDREQ count on gfx950 for the two access patterns of
// k_extend6: a coalesced 16-byte-per-lane stream (the rays) and a per-lane gather of 64-byte records
// (node-pair / leaf records, four dwordx4 loads per lane).
//
// DEVELOPER TOOL (tests/tools).  Run under rocprofv3 --pmc FETCH_SIZE (and, in another pass,
// TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum); every dispatch prints the bytes it
// requested, so counter / bytes gives the correction factor per pattern and residency:
//   stream_1g    1 GiB read once, first touch after a 512 MiB flush      -> HBM
//   stream_64m   64 MiB read, warm (second pass)                          -> Infinity Cache resident
//   gather_S     2 M lanes x 16 records of 64 B drawn uniformly from a table of S bytes
//                (S = 6 MB: L2 resident; 96 MB: Infinity Cache; 1.5 GB: HBM)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void calib_stream_read(const float4* __restrict__ src, size_t n16, float* sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void calib_gather64(const float4* __restrict__ table, unsigned nrec, int per_lane, float* sink)
{
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int k = 0; k < per_lane; ++k) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const float4* r = table + (size_t)(s % nrec) * 4;
        const float4 a = r[0], b = r[1], c = r[2], d = r[3];
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void calib_flush_write(float4* dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        dst[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

// L1 (TCP) tag-lookup rate: every lane reads 16 B at its own pseudo-random 16-byte slot of a small table,
// `per_lane` dependent-free loads per lane (8 in flight); lookups per clock per CU = lanes x loads / cycles.
__global__ __launch_bounds__(256) void calib_l1_lookups(const float4* __restrict__ table, unsigned nslots, int per_lane, float* sink)
{
    unsigned s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 777u;
    float acc = 0.f;
    for (int k = 0; k < per_lane; k += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            v[j] = table[s % nslots];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j].x;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static void time_l1(float4* big, float* sink)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int per_lane = 1024;
    for (size_t bytes : {(size_t)8 << 10, (size_t)16 << 10, (size_t)256 << 10, (size_t)2 << 20, (size_t)6 << 20}) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(calib_l1_lookups, dim3(256 * 8), dim3(256), 0, nullptr, big, (unsigned)(bytes / 16), per_lane, sink);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double lookups = 256.0 * 8 * 256 * per_lane;
        printf("l1_lookups table %7zu KB: %.3f ms, %.2f lane-lookups per ns chip-wide = %.3f per CU per clock at 2.4 GHz\n",
               bytes >> 10, best, lookups / (best * 1e6), lookups / (best * 1e-3 * 2.4e9 * 256));
    }
}

int main(int argc, char** argv)
{
    if (argc > 1 && argv[1][0] == 'l') {      // `fetch_calib l1`: timing only, no counters needed
        float* sink0;
        float4* tab;
        CK(hipMalloc(&sink0, 64));
        CK(hipMalloc(&tab, (size_t)8 << 20));
        CK(hipMemset(tab, 0, (size_t)8 << 20));
        time_l1(tab, sink0);
        return 0;
    }
    float* sink;
    CK(hipMalloc(&sink, 64));
    const size_t G = (size_t)1 << 30;
    float4 *big, *flush;
    CK(hipMalloc(&big, G + (G >> 1)));
    CK(hipMalloc(&flush, G >> 1));
    CK(hipMemset(big, 0, G + (G >> 1)));
    auto do_flush = [&] { hipLaunchKernelGGL(calib_flush_write, dim3(4096), dim3(256), 0, nullptr, flush, (G >> 1) / 16); CK(hipDeviceSynchronize()); };

    do_flush();
    hipLaunchKernelGGL(calib_stream_read, dim3(8192), dim3(256), 0, nullptr, big, G / 16, sink);
    CK(hipDeviceSynchronize());
    printf("dispatch stream_1g   calib_stream_read grid 8192 bytes %zu\n", G);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(calib_stream_read, dim3(4096), dim3(256), 0, nullptr, big, ((size_t)64 << 20) / 16, sink);
        CK(hipDeviceSynchronize());
        printf("dispatch stream_64m%s calib_stream_read grid 4096 bytes %zu\n", rep ? "_warm" : "_cold", (size_t)64 << 20);
    }
    const size_t sizes[3] = {(size_t)6 << 20, (size_t)96 << 20, (size_t)3 << 29};
    const unsigned grids[3] = {8100, 8101, 8102};     // distinct grid sizes identify the rows in the CSV
    for (int k = 0; k < 3; ++k) {
        do_flush();
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(calib_gather64, dim3(grids[k]), dim3(256), 0, nullptr, big, (unsigned)(sizes[k] / 64), 16, sink);
            CK(hipDeviceSynchronize());
            printf("dispatch gather_%zuMB_%s calib_gather64 grid %u bytes %zu (requested; distinct records <= table %zu)\n",
                   sizes[k] >> 20, rep ? "warm" : "cold", grids[k], (size_t)grids[k] * 256 * 16 * 64, sizes[k]);
        }
    }
    return 0;
}
