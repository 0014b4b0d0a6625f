// atomic_calib.hip -- DEVELOPER TOOL (tests/tools): how many scattered int32 atomic adds per second the MI355X's memory side
// retires, in the shape of k_extend6's deposits (extend.cl:94-98): every lane adds 1 to a pseudo-random counter of a table of
// REPLICAS x T ints (16 x 44 866 = the context's count planes, 2.9 MB: L2-resident), no return value, 64 lanes -> ~64 different
// lines per wave-instruction.  The kernel's deposits all leave L2 (TCC_EA0_ATOMIC = hits), so this rate is a unit of its own in
// the bench line's roofline.
//   hipcc --offload-arch=gfx950 -O3 -o tests/tools/atomic_calib tests/tools/atomic_calib.hip && tests/tools/atomic_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool HOT>
__global__ __launch_bounds__(256) void k_atomics(int32_t* table, uint32_t n, int per_lane, uint32_t hot_mask)
{
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int k = 0; k < per_lane; ++k) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        uint32_t idx = (uint32_t)(((uint64_t)s * n) >> 32);
        if (HOT) idx &= hot_mask;                 // a few hot counters: the lamp's nearest triangles take 1-2 % of all hits each
        __hip_atomic_fetch_add(&table[idx], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // result unused: no-return form
    }
}

int main()
{
    const uint32_t T = 44866, replicas = 16, n = T * replicas;
    int32_t* table;
    CK(hipMalloc(&table, (size_t)n * 4));
    CK(hipMemset(table, 0, (size_t)n * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("table %u ints (%u replicas x %u), scattered no-return int32 atomic adds\n", n, replicas, T);
    for (int per_cu : {8, 4, 2, 1}) {
        const unsigned grid = 256u * per_cu;
        const int per_lane = 256;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_atomics<false>, dim3(grid), dim3(256), 0, nullptr, table, n, per_lane, 0u);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double total = (double)grid * 256 * per_lane;
        printf("  %d workgroups per CU: %8.3f ms for %.0f M atomics = %7.2f G atomics/s\n", per_cu, best, total / 1e6, total / (best * 1e6));
    }
    {   // one address per replica group: the contention limit
        const unsigned grid = 256u * 8;
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(k_atomics<true>, dim3(grid), dim3(256), 0, nullptr, table, n, 64, 15u);
            CK(hipEventRecord(e1, nullptr));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double total = (double)grid * 256 * 64;
        printf("  16 counters only (contended): %8.3f ms = %7.2f G atomics/s\n", best, total / (best * 1e6));
    }
    return 0;
}
