// rcp_probe.hip -- reads gfx950's v_rcp_f32 off the GPU for the oracle's "shipped flags" flavour.
//
// TEST INFRASTRUCTURE ONLY (linked into oracle/libref_gpu.so; tests/ and bench.py's reporting leg).
// refgpu_rcp_table: the instruction's result for every 23-bit significand of [1, 2) -- the table
// oracle/rcp_model.h turns into a model of the instruction over all of binary32.
// refgpu_rcp_check: that model (the same header, compiled for the device) against the instruction on ALL 2^32
// inputs; returns the number of mismatches and the first few of them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rcp_model.h"

namespace {

__device__ __forceinline__ uint32_t hw_rcp_bits(uint32_t x)
{
    float y;
    asm volatile("v_rcp_f32 %0, %1" : "=v"(y) : "v"(__uint_as_float(x)));
    return __float_as_uint(y);
}

__global__ __launch_bounds__(256) void k_rcp_table(uint32_t* table)
{
    const uint32_t m = blockIdx.x * 256u + threadIdx.x;           // grid = 2^23 / 256
    table[m] = hw_rcp_bits(0x3F800000u | m);
}

// grid-stride over all 2^32 bit patterns
__global__ __launch_bounds__(256) void k_rcp_check(const uint32_t* table, unsigned long long* nbad, uint32_t* bad3, uint32_t bad_cap)
{
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    unsigned long long mine = 0;
    for (uint64_t x = (uint64_t)blockIdx.x * 256u + threadIdx.x; x < (1ull << 32); x += stride) {
        const uint32_t hw = hw_rcp_bits((uint32_t)x);
        const uint32_t md = orc_rcp_model_bits((uint32_t)x, table);
        const bool nan_both = ((hw & 0x7F800000u) == 0x7F800000u && (hw & 0x007FFFFFu)) &&
                              ((md & 0x7F800000u) == 0x7F800000u && (md & 0x007FFFFFu));
        if (hw != md && !nan_both) {
            ++mine;
            const unsigned long long slot = atomicAdd(nbad + 1, 1ull);
            if (slot < bad_cap) { bad3[slot * 3] = (uint32_t)x; bad3[slot * 3 + 1] = hw; bad3[slot * 3 + 2] = md; }
        }
    }
    if (mine) atomicAdd(nbad, mine);
}

}  // namespace

extern "C" {

// table: 2^23 uint32 on the host
int refgpu_rcp_table(uint32_t* table)
{
    uint32_t* d = nullptr;
    if (hipMalloc(&d, (size_t)4 << 23) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_rcp_table, dim3(1u << 15), dim3(256), 0, nullptr, d);
    if (hipMemcpy(table, d, (size_t)4 << 23, hipMemcpyDeviceToHost) != hipSuccess) { hipFree(d); return -1; }
    hipFree(d);
    return 0;
}

// the model built from `table` against the instruction over all 2^32 inputs: *mismatches = how many differ (NaN
// results compare equal whatever their payload); bad3 receives up to bad_cap triples (input, hardware, model)
int refgpu_rcp_check(const uint32_t* table, uint64_t* mismatches, uint32_t* bad3, uint32_t bad_cap)
{
    uint32_t *d = nullptr, *dbad = nullptr;
    unsigned long long* dn = nullptr;
    if (hipMalloc(&d, (size_t)4 << 23) != hipSuccess) return -1;
    if (hipMalloc(&dn, 16) != hipSuccess) return -1;
    if (hipMalloc(&dbad, (size_t)(bad_cap ? bad_cap : 1) * 12) != hipSuccess) return -1;
    hipMemcpy(d, table, (size_t)4 << 23, hipMemcpyHostToDevice);
    hipMemset(dn, 0, 16);
    hipMemset(dbad, 0, (size_t)(bad_cap ? bad_cap : 1) * 12);
    hipLaunchKernelGGL(k_rcp_check, dim3(256 * 32), dim3(256), 0, nullptr, d, dn, dbad, bad_cap);
    unsigned long long n[2] = {0, 0};
    if (hipMemcpy(n, dn, 16, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    if (bad_cap) hipMemcpy(bad3, dbad, (size_t)bad_cap * 12, hipMemcpyDeviceToHost);
    *mismatches = n[0];
    hipFree(d); hipFree(dn); hipFree(dbad);
    return 0;
}

}  // extern "C"
