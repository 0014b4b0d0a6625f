// Exhaustive check of the packed-division identity used by extend v5 (csrc/uvrt_extend5.hip):
//
//     y  = RN32( RN64( 1 / (double)d ) )
//     q0 = RN32(a * y);   r = fma(-d, q0, a);   q = fma(r, y, q0)        ==>   q == RN32(a / d)
//
// for ALL pairs of binary32 significands: a = A, d = D with A, D in [2^23, 2^24).  The three steps
// are exact scalings under a change of either exponent (no overflow / underflow: the kernel sends
// lanes and scenes outside that range to the IEEE division), and signs factor out, so the 2^46
// pairs cover every normal operand pair.  ~40 s on one MI355X.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
//         tests/tools/div3_exhaustive.hip -o /tmp/div3 && /tmp/div3 [stride]
//
// stride > 1 checks every stride-th divisor only (the GPU test uses 64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__global__ __launch_bounds__(256) void k_check(uint32_t d_first, uint32_t d_stride, uint32_t a_chunks,
                                               unsigned long long* bad, uint32_t* first_bad)
{
    // one thread: one divisor, 2^23 / a_chunks consecutive dividends
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const uint32_t di = t / a_chunks, ac = t % a_chunks;
    const uint32_t D = d_first + di * d_stride;
    if (D >= (1u << 24)) return;
    const float d = (float)D;
    const float y = (float)(1.0 / (double)d);
    const uint32_t per = (1u << 23) / a_chunks;
    uint32_t A = (1u << 23) + ac * per;
    unsigned long long nbad = 0;
    for (uint32_t i = 0; i < per; ++i, ++A) {
        const float a = (float)A;
        const float q0 = a * y;
        const float r = __builtin_fmaf(-d, q0, a);
        const float q = __builtin_fmaf(r, y, q0);
        const float ref = a / d;
        if (__float_as_uint(q) != __float_as_uint(ref)) {
            if (nbad == 0) { first_bad[0] = A; first_bad[1] = D; }
            ++nbad;
        }
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main(int argc, char** argv)
{
    const uint32_t stride = argc > 1 ? (uint32_t)atoi(argv[1]) : 1u;
    unsigned long long* bad; uint32_t* first_bad;
    hipMalloc(&bad, 8); hipMalloc(&first_bad, 8);
    hipMemset(bad, 0, 8); hipMemset(first_bad, 0, 8);
    const uint32_t a_chunks = 64;                       // 131072 dividends per thread
    const uint32_t d_per_launch = 1u << 17;             // divisors per launch
    unsigned long long pairs = 0;
    for (uint32_t base = 1u << 23; base < (1u << 24); base += d_per_launch * stride) {
        const uint32_t threads = d_per_launch * a_chunks;
        hipLaunchKernelGGL(k_check, dim3(threads / 256), dim3(256), 0, 0, base, stride, a_chunks, bad, first_bad);
        if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 2; }
        pairs += (unsigned long long)d_per_launch * (1ull << 23);
        unsigned long long h = 0;
        hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
        printf("divisors [%u, +%u step %u): %llu pairs so far, mismatches %llu\n", base, d_per_launch * stride, stride, pairs, h);
        fflush(stdout);
    }
    unsigned long long h = 0; uint32_t fb[2];
    hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(fb, first_bad, 8, hipMemcpyDeviceToHost);
    printf("TOTAL pairs %llu  mismatches %llu", pairs, h);
    if (h) printf("  (e.g. A=%u D=%u)", fb[0], fb[1]);
    printf("\n");
    return h ? 1 : 0;
}
