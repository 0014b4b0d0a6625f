// gather_coop.hip -- DEVELOPER TOOL (tests/tools): how fast can a wave follow per-lane chains of dependent 64-byte record
// fetches from a table far beyond the caches (the access pattern of k_extend6 on a scene of millions of triangles), by
// the form of the fetch:
//   lane4   every lane loads its own record with four global_load_dwordx4 (what run7 does): a wave-instruction touches 64
//           different lines for 16 bytes each -- 256 L1 tag lookups per wave and record
//   coop    the lanes of a wave fetch each other's records: instruction k (of 4) brings the records of lanes 16k..16k+15,
//           lane l the 16-byte piece (l & 3) of the record of lane 16k + (l >> 2) -- four neighbouring lanes read 64
//           contiguous bytes, 64 lookups per wave and record -- straight into LDS (global_load_lds_dwordx4: destination =
//           wave-uniform base + lane x 16, so lane r's record lands at base + 64 r), then four ds_read_b128 per lane
//   coopx   the same with the pieces of a record XOR-swizzled by (lane >> 2) & 3 on the source side, so that the
//           ds_read_b128 of 16 neighbouring lanes hit 16 different bank groups
// Every lane follows STEPS dependent fetches (next index = a hash of the words it read); the grid is sized by
// workgroups per CU (occupancy is what the real kernel would have with the staging area in LDS).
//   hipcc --offload-arch=gfx950 -O3 -o tests/tools/gather_coop tests/tools/gather_coop.hip && tests/tools/gather_coop
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void fill_random(uint4* t, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        uint32_t s = (uint32_t)i * 2654435761u + 99991u;
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        const uint32_t a = s;
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        t[i] = make_uint4(a, s, a ^ 0x5bd1e995u, s + 0x9e3779b9u);
    }
}

// the next record: with probability hot16 / 16 one of the first `nhot` records (a part of the table that stays in every
// XCD's L2: the upper levels of a tree), else any record of the table
__device__ uint32_t g_nhot, g_hot16;
__device__ __forceinline__ uint32_t next_index(uint4 a, uint4 b, uint4 c, uint4 d, uint32_t nrec)
{
    const uint32_t h = (a.x ^ b.y ^ c.z ^ d.w) * 2654435761u;
    const uint32_t h2 = (h ^ (h >> 15)) * 0x2c1b3c6du;
    const uint32_t range = (h2 >> 28) < g_hot16 ? g_nhot : nrec;
    return (uint32_t)(((uint64_t)h * range) >> 32);
}

template <int STEPS>
__global__ __launch_bounds__(256) void chain_lane4(const uint4* __restrict__ table, uint32_t nrec, uint32_t* sink, int pad)
{
    extern __shared__ uint4 s_pad[];          // occupancy control only
    if (pad < 0) s_pad[threadIdx.x] = make_uint4(0, 0, 0, 0);
    uint32_t idx = (uint32_t)(((uint64_t)((blockIdx.x * 256u + threadIdx.x) * 2654435761u) * nrec) >> 32);
    uint32_t acc = 0;
    for (int k = 0; k < STEPS; ++k) {
        const uint4* r = table + (size_t)idx * 4;
        const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
        idx = next_index(a, b, c, d, nrec);
        acc += idx;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int STEPS, bool SWZ>
__global__ __launch_bounds__(256) void chain_coop(const uint4* __restrict__ table, uint32_t nrec, uint32_t* sink, int pad)
{
    extern __shared__ uint4 s_stage[];        // [4 waves][64 records][4 pieces] = 16 KB (+ pad for occupancy control)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint4* const my_stage = s_stage + wave * 256;                      // wave-uniform
    uint32_t idx = (uint32_t)(((uint64_t)((blockIdx.x * 256u + threadIdx.x) * 2654435761u) * nrec) >> 32);
    uint32_t acc = 0;
    const uint32_t piece = lane & 3u;
    for (int k = 0; k < STEPS; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t owner = 16u * j + (lane >> 2);                                    // the lane whose record this is
            const uint32_t oidx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)idx);
            const uint32_t pc = SWZ ? (piece ^ ((owner >> 2) & 3u)) : piece;
            const uint4* src = table + (size_t)oidx * 4 + pc;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                             (void __attribute__((address_space(3)))*)(my_stage + 64 * j), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint4* rec = my_stage + lane * 4;
        const uint32_t f = SWZ ? ((lane >> 2) & 3u) : 0u;
        const uint4 a = rec[0 ^ f], b = rec[1 ^ f], c = rec[2 ^ f], d = rec[3 ^ f];
        idx = next_index(a, b, c, d, nrec);
        acc += idx;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the reads are done before the next fill overwrites
    }
    if (acc == 0x12345678u || pad < 0) sink[0] = acc;
}

int main(int argc, char** argv)
{
    const size_t mb = argc > 1 ? (size_t)atol(argv[1]) : 768;
    const uint32_t hot16 = argc > 2 ? (uint32_t)atoi(argv[2]) : 0;                 // sixteenths of the fetches that go to the hot part
    const uint32_t nhot = (argc > 3 ? (uint32_t)atoi(argv[3]) : 2048u) * 1024u / 64u;  // its size in KB
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_nhot), &nhot, 4));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_hot16), &hot16, 4));
    const size_t bytes = mb << 20;
    const uint32_t nrec = (uint32_t)(bytes / 64);
    uint4* table;
    uint32_t* sink;
    CK(hipMalloc(&table, bytes));
    CK(hipMalloc(&sink, 64));
    hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, nullptr, table, bytes / 16);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    constexpr int STEPS = 64;
    const int waves_total = 256 * 8 * 4 * 8;                          // lanes x STEPS records per launch
    const unsigned grid = waves_total / 4;
    printf("table %zu MB (%u records of 64 B), %u/16 of the fetches from its first %u KB, %d dependent fetches per lane, %u workgroups of 256\n",
           mb, nrec, hot16, nhot * 64 / 1024, STEPS, grid);
    for (int per_cu : {8, 6, 4}) {
        // dynamic LDS so that exactly per_cu workgroups fit a CU's 160 KB
        const size_t lds = ((size_t)160 * 1024 / per_cu) & ~(size_t)255;
        for (int form = 0; form < 3; ++form) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, nullptr));
                if (form == 0) hipLaunchKernelGGL(chain_lane4<STEPS>, dim3(grid), dim3(256), lds, nullptr, table, nrec, sink, 0);
                else if (form == 1) hipLaunchKernelGGL((chain_coop<STEPS, false>), dim3(grid), dim3(256), lds, nullptr, table, nrec, sink, 0);
                else hipLaunchKernelGGL((chain_coop<STEPS, true>), dim3(grid), dim3(256), lds, nullptr, table, nrec, sink, 0);
                CK(hipEventRecord(e1, nullptr));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double recs = (double)grid * 256 * STEPS;
            printf("  %d workgroups per CU  %-6s %8.3f ms  %7.1f G records/s  %6.2f TB/s\n", per_cu,
                   form == 0 ? "lane4" : form == 1 ? "coop" : "coopx", best, recs / (best * 1e6), recs * 64 / (best * 1e9));
        }
    }
    return 0;
}
