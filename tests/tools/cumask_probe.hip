// Developer probe: does a HIP stream created with a CU mask (hipExtStreamCreateWithCUMask) keep its kernels off the masked-out
// compute units on the MI355X (8 XCDs, 256 CUs), and which mask bit is which CU?  A census kernel records where each workgroup
// ran (XCC id; shader engine / array / CU id from HW_REG_HW_ID); the host prints the CUs an unmasked stream uses, the CU that
// disappears when ONE mask bit is cleared (for a sample of bits), and what masks without the last / first N bits leave.
// Build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_census(unsigned* out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // keep the workgroup resident for a while so that the grid spreads over every CU the stream may use
    unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

// CU key: xcc [19:16], se_id [15:13], sh_id [12], cu_id [11:8] of HW_REG_HW_ID
static int census(hipStream_t s, unsigned* d_out, int blocks, std::set<unsigned>& cus)
{
    std::vector<unsigned> h(2 * blocks);
    hipLaunchKernelGGL(k_census, dim3(blocks), dim3(256), 0, s, d_out, 100000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
    cus.clear();
    for (int b = 0; b < blocks; ++b) cus.insert(((h[2 * b + 1] & 0xF) << 16) | (h[2 * b] & 0xFF00));
    return 0;
}

static void per_xcc(const std::set<unsigned>& cus, char* buf, size_t n)
{
    int cnt[16] = {};
    for (unsigned c : cus) ++cnt[c >> 16];
    snprintf(buf, n, "%d %d %d %d %d %d %d %d", cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7]);
}

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("%s, %d CUs\n", prop.gcnArchName, ncu);
    const int blocks = 4096;
    unsigned* d_out;
    CK(hipMalloc(&d_out, 2 * blocks * 4));
    hipStream_t plain;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    std::set<unsigned> all, got;
    if (census(plain, d_out, blocks, all)) return 1;
    char buf[128];
    per_xcc(all, buf, sizeof buf);
    printf("unmasked stream: %zu distinct CUs, per XCC: %s\n", all.size(), buf);
    const int words = (ncu + 31) / 32;
    auto run_mask = [&](const std::vector<unsigned>& mask, std::set<unsigned>& out) -> int {
        hipStream_t ms;
        hipError_t e = hipExtStreamCreateWithCUMask(&ms, (uint32_t)words, mask.data());
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); return 1; }
        const int rc = census(ms, d_out, blocks, out);
        (void)hipStreamDestroy(ms);
        return rc;
    };
    // one bit cleared: which CU disappears?
    for (int bit : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 254, 255}) {
        if (bit >= ncu) continue;
        std::vector<unsigned> mask(words, 0xFFFFFFFFu);
        mask[bit / 32] &= ~(1u << (bit % 32));
        if (run_mask(mask, got)) return 1;
        printf("bit %3d cleared: %3zu CUs used; missing:", bit, got.size());
        for (unsigned c : all) if (!got.count(c)) printf(" xcc %u se %u sh %u cu %u", c >> 16, (c >> 13) & 7, (c >> 12) & 1, (c >> 8) & 15);
        printf("\n");
    }
    // the first / last N bits cleared
    for (int n : {8, 16, 32}) {
        for (int last = 0; last < 2; ++last) {
            std::vector<unsigned> mask(words, 0xFFFFFFFFu);
            for (int k = 0; k < n; ++k) { const int bit = last ? ncu - 1 - k : k; mask[bit / 32] &= ~(1u << (bit % 32)); }
            if (run_mask(mask, got)) return 1;
            per_xcc(got, buf, sizeof buf);
            printf("%s %2d bits cleared: %3zu CUs used, per XCC: %s\n", last ? "last " : "first", n, got.size(), buf);
        }
    }
    // bits 0, 8, 16, ..., (every 8th): the round-2 probe saw no CU disappear for this one
    {
        std::vector<unsigned> mask(words, 0xFFFFFFFFu);
        for (int bit = 0; bit < ncu; bit += 8) mask[bit / 32] &= ~(1u << (bit % 32));
        if (run_mask(mask, got)) return 1;
        per_xcc(got, buf, sizeof buf);
        printf("every 8th bit cleared: %3zu CUs used, per XCC: %s\n", got.size(), buf);
    }
    return 0;
}
