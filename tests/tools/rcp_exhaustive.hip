// Exhaustive check of the reciprocal used by extend v6's triangle test (f = 1.0f / a,
// extend.cl:17):   y0 = v_rcp_f32(a); e = fma(-a, y0, 1); y1 = fma(e, y0, y0);
//                  r = fma(-a, y1, 1); y = fma(r, y1, y1)          ==>  y == RN32(1 / a)
// for every binary32 a with 2^-64 <= |a| < 2^64 (the kernel guarantees |a| >= 1e-5 and sends
// anything larger than 2^60 to the IEEE division).  v_rcp_f32 is a hardware approximation, so this
// can only be checked on the GPU itself.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ float rcp5(float a)
{
    float y0;
    asm("v_rcp_f32 %0, %1" : "=v"(y0) : "v"(a));
    const float e = __builtin_fmaf(-a, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float r = __builtin_fmaf(-a, y1, 1.0f);
    return __builtin_fmaf(r, y1, y1);
}
__device__ __forceinline__ float rcp3(float a)
{
    float y0;
    asm("v_rcp_f32 %0, %1" : "=v"(y0) : "v"(a));
    const float e = __builtin_fmaf(-a, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}

__global__ __launch_bounds__(256) void k_check(uint32_t exp_first, unsigned long long* bad, uint32_t* first_bad)
{
    // one block row per biased exponent, one thread per 2^23/… mantissas
    const uint32_t e = exp_first + blockIdx.y;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;        // 0 .. 2^15
    unsigned long long nb5 = 0, nb3 = 0;
    for (uint32_t i = 0; i < 256; ++i) {
        const uint32_t m = t * 256 + i;
        for (uint32_t s = 0; s < 2; ++s) {
            const float a = __uint_as_float((s << 31) | (e << 23) | m);
            const float ref = 1.0f / a;
            if (__float_as_uint(rcp5(a)) != __float_as_uint(ref)) { if (!nb5) { first_bad[0] = __float_as_uint(a); } ++nb5; }
            if (__float_as_uint(rcp3(a)) != __float_as_uint(ref)) ++nb3;
        }
    }
    if (nb5) atomicAdd(&bad[0], nb5);
    if (nb3) atomicAdd(&bad[1], nb3);
}

int main()
{
    unsigned long long* bad; uint32_t* first_bad;
    hipMalloc(&bad, 16); hipMalloc(&first_bad, 8);
    hipMemset(bad, 0, 16); hipMemset(first_bad, 0, 8);
    const uint32_t e0 = 127 - 64, ne = 128;
    hipLaunchKernelGGL(k_check, dim3(128, ne), dim3(256), 0, 0, e0, bad, first_bad);
    if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 2; }
    unsigned long long h[2]; uint32_t fb[2];
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
    hipMemcpy(fb, first_bad, 8, hipMemcpyDeviceToHost);
    printf("values %llu  mismatches: 5-op %llu, 3-op %llu", 2ull * ne * (1ull << 23), h[0], h[1]);
    if (h[0]) printf("  (e.g. a bits 0x%08x)", fb[0]);
    printf("\n");
    return h[0] ? 1 : 0;
}
