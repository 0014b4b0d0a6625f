// valu_calib.hip -- issue-rate calibration of the instruction classes k_extend6 is made of (gfx950).
//
// DEVELOPER TOOL (tests/tools): not part of the product, not linked by it.  Build + run:
//     hipcc --offload-arch=gfx950 -O2 -o valu_calib tests/tools/valu_calib.hip && ./valu_calib
// Under rocprofv3 --pmc it also calibrates what SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU_* count per
// instruction (each class runs as its own kernel, named after the class).
//
// Every kernel: 256-thread workgroups, `wgs_per_cu` of them per CU (1, 2, 4, 8 waves per SIMD), each
// wave runs ITER iterations of a block of 64 independent instructions of one class (8 accumulators x 8)
// and stamps s_memtime around the loop.  Reported: shader cycles per wave-instruction per SIMD =
// median wave's cycles / (instructions per wave x waves per SIMD), and the same from wall time at the
// measured clock.  The `exec` variants run the class with part of the lanes masked off.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int ITER = 512;
constexpr int PER_BLOCK = 64;

struct Out { unsigned long long cycles; unsigned long long realtime; };

#define REP8(S) S S S S S S S S

#define BODY_SCALAR(INSN)                                                                      \
    asm volatile(REP8(INSN " %0, %8, %9, %0\n\t" INSN " %1, %8, %9, %1\n\t" INSN " %2, %8, %9, %2\n\t" \
                      INSN " %3, %8, %9, %3\n\t" INSN " %4, %8, %9, %4\n\t" INSN " %5, %8, %9, %5\n\t" \
                      INSN " %6, %8, %9, %6\n\t" INSN " %7, %8, %9, %7\n\t")                             \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "v"(x), "v"(y))

#define BODY_2OP(INSN)                                                                         \
    asm volatile(REP8(INSN " %0, %8, %0\n\t" INSN " %1, %8, %1\n\t" INSN " %2, %8, %2\n\t"            \
                      INSN " %3, %8, %3\n\t" INSN " %4, %8, %4\n\t" INSN " %5, %8, %5\n\t"            \
                      INSN " %6, %8, %6\n\t" INSN " %7, %8, %7\n\t")                                  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "v"(x))

#define BODY_1OP(INSN)                                                                         \
    asm volatile(REP8(INSN " %0, %0\n\t" INSN " %1, %1\n\t" INSN " %2, %2\n\t" INSN " %3, %3\n\t"     \
                      INSN " %4, %4\n\t" INSN " %5, %5\n\t" INSN " %6, %6\n\t" INSN " %7, %7\n\t")    \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))

// exec_mode: 0 all lanes, 1 lanes 0-31, 2 lanes 32-63, 3 even lanes, 4 lane 0 only, 5 lanes 0-15
__device__ __forceinline__ void set_exec(int exec_mode)
{
    if (exec_mode == 1) asm volatile("s_mov_b64 exec, 0x00000000ffffffff");
    else if (exec_mode == 2) asm volatile("s_mov_b32 exec_lo, 0\n\ts_mov_b32 exec_hi, -1");
    else if (exec_mode == 3) asm volatile("s_mov_b32 exec_lo, 0x55555555\n\ts_mov_b32 exec_hi, 0x55555555");
    else if (exec_mode == 4) asm volatile("s_mov_b64 exec, 1");
    else if (exec_mode == 5) asm volatile("s_mov_b64 exec, 0xffff");
}

#define KERNEL_PROLOGUE                                                                         \
    float a0 = seed[threadIdx.x & 7], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f,   \
          a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;                                            \
    const float x = seed[8], y = seed[9];                                                        \
    (void)y;                                                                                    \
    __syncthreads();                                                                            \
    unsigned long long t0, t1, r0, r1;                                                          \
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0), "=s"(t0) :: "memory");

#define KERNEL_EPILOGUE                                                                         \
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory"); \
    asm volatile("s_mov_b64 exec, -1");                                                         \
    if ((threadIdx.x & 63) == 0) {                                                              \
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);                                      \
        out[w].cycles = t1 - t0;                                                                \
        out[w].realtime = r1 - r0;                                                              \
    }                                                                                           \
    sink[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;

#define DEF_KERNEL(NAME, BODY)                                                                  \
    __global__ __launch_bounds__(256) void NAME(const float* seed, float* sink, Out* out, int exec_mode) \
    {                                                                                           \
        KERNEL_PROLOGUE                                                                         \
        set_exec(exec_mode);                                                                    \
        for (int i = 0; i < ITER; ++i) { BODY; }                                                \
        KERNEL_EPILOGUE                                                                         \
    }

DEF_KERNEL(calib_v_fma_f32, BODY_SCALAR("v_fma_f32"))
DEF_KERNEL(calib_v_mul_f32, BODY_2OP("v_mul_f32"))
DEF_KERNEL(calib_v_add_f32, BODY_2OP("v_add_f32"))
DEF_KERNEL(calib_v_max_f32, BODY_2OP("v_max_f32"))
DEF_KERNEL(calib_v_min3_f32, BODY_SCALAR("v_min3_f32"))
DEF_KERNEL(calib_v_rcp_f32, BODY_1OP("v_rcp_f32"))
DEF_KERNEL(calib_v_mov_b32, BODY_1OP("v_mov_b32"))
DEF_KERNEL(calib_v_and_b32, BODY_2OP("v_and_b32"))
DEF_KERNEL(calib_v_lshl_add_u32, BODY_SCALAR("v_lshl_add_u32"))
#define BODY_CND                                                                                \
    asm volatile(REP8("v_cndmask_b32 %0, %8, %0, vcc\n\t v_cndmask_b32 %1, %8, %1, vcc\n\t v_cndmask_b32 %2, %8, %2, vcc\n\t" \
                      "v_cndmask_b32 %3, %8, %3, vcc\n\t v_cndmask_b32 %4, %8, %4, vcc\n\t v_cndmask_b32 %5, %8, %5, vcc\n\t" \
                      "v_cndmask_b32 %6, %8, %6, vcc\n\t v_cndmask_b32 %7, %8, %7, vcc\n\t")                                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "v"(x))
DEF_KERNEL(calib_v_cndmask_b32, BODY_CND)      // VOP2: vcc as the selector

DEF_KERNEL(calib_v_sub_f32, BODY_2OP("v_sub_f32"))
DEF_KERNEL(calib_v_min_f32, BODY_2OP("v_min_f32"))
DEF_KERNEL(calib_v_or_b32, BODY_2OP("v_or_b32"))
DEF_KERNEL(calib_v_xor_b32, BODY_2OP("v_xor_b32"))
DEF_KERNEL(calib_v_lshlrev_b32, BODY_2OP("v_lshlrev_b32"))
DEF_KERNEL(calib_v_lshrrev_b32, BODY_2OP("v_lshrrev_b32"))
DEF_KERNEL(calib_v_add_u32, BODY_2OP("v_add_u32"))
DEF_KERNEL(calib_v_sub_u32, BODY_2OP("v_sub_u32"))
DEF_KERNEL(calib_v_mul_lo_u32, BODY_2OP("v_mul_lo_u32"))
DEF_KERNEL(calib_v_mul_u32_u24, BODY_2OP("v_mul_u32_u24"))
DEF_KERNEL(calib_v_mad_u32_u24, BODY_SCALAR("v_mad_u32_u24"))
DEF_KERNEL(calib_v_bfe_u32, BODY_SCALAR("v_bfe_u32"))
DEF_KERNEL(calib_v_bfi_b32, BODY_SCALAR("v_bfi_b32"))
DEF_KERNEL(calib_v_and_or_b32, BODY_SCALAR("v_and_or_b32"))
DEF_KERNEL(calib_v_add3_u32, BODY_SCALAR("v_add3_u32"))
DEF_KERNEL(calib_v_med3_f32, BODY_SCALAR("v_med3_f32"))
DEF_KERNEL(calib_v_mul_f32_e64, BODY_2OP("v_mul_f32_e64"))
DEF_KERNEL(calib_v_max_f32_e64, BODY_2OP("v_max_f32_e64"))
DEF_KERNEL(calib_v_fmac_f32, BODY_2OP("v_fmac_f32"))       // dst += x * dst
DEF_KERNEL(calib_v_mul_legacy_f32, BODY_2OP("v_mul_legacy_f32"))
// a VALU instruction with one SGPR source operand
#define BODY_SGPR_SRC                                                                           \
    asm volatile(REP8("v_add_f32 %0, %8, %0\n\t v_add_f32 %1, %8, %1\n\t v_add_f32 %2, %8, %2\n\t v_add_f32 %3, %8, %3\n\t" \
                      "v_add_f32 %4, %8, %4\n\t v_add_f32 %5, %8, %5\n\t v_add_f32 %6, %8, %6\n\t v_add_f32 %7, %8, %7\n\t") \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "s"(sx))
// v_cndmask with an SGPR-pair selector (VOP3)
#define BODY_CND_SGPR                                                                           \
    asm volatile(REP8("v_cndmask_b32 %0, %8, %0, %9\n\t v_cndmask_b32 %1, %8, %1, %9\n\t v_cndmask_b32 %2, %8, %2, %9\n\t" \
                      "v_cndmask_b32 %3, %8, %3, %9\n\t v_cndmask_b32 %4, %8, %4, %9\n\t v_cndmask_b32 %5, %8, %5, %9\n\t" \
                      "v_cndmask_b32 %6, %8, %6, %9\n\t v_cndmask_b32 %7, %8, %7, %9\n\t")                                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "v"(x), "s"(smask))
// compare + select pairs, as a compiled `c ? a : b` looks: v_cmp -> vcc, then v_cndmask reads vcc
#define BODY_CMP_CND                                                                            \
    asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %8\n\t v_cndmask_b32 %1, %8, %1, vcc\n\t v_cmp_lt_f32 vcc, %2, %8\n\t v_cndmask_b32 %3, %8, %3, vcc\n\t" \
                      "v_cmp_lt_f32 vcc, %4, %8\n\t v_cndmask_b32 %5, %8, %5, vcc\n\t v_cmp_lt_f32 vcc, %6, %8\n\t v_cndmask_b32 %7, %8, %7, vcc\n\t") \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "v"(x) : "vcc")
// v_cmp (VOP3) writing an SGPR pair other than vcc
#define BODY_CMP_SGPR                                                                           \
    asm volatile(REP8("v_cmp_lt_f32 %8, %0, %1\n\t v_cmp_lt_f32 %9, %1, %2\n\t v_cmp_lt_f32 %8, %2, %3\n\t v_cmp_lt_f32 %9, %3, %4\n\t" \
                      "v_cmp_lt_f32 %8, %4, %5\n\t v_cmp_lt_f32 %9, %5, %6\n\t v_cmp_lt_f32 %8, %6, %7\n\t v_cmp_lt_f32 %9, %7, %0\n\t") \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(sm0), "=&s"(sm1))
#define BODY_1OP_PAIRLESS BODY_1OP("v_mov_b32")
#define SGPR_PROLOGUE                                                                           \
    const float sx = __builtin_bit_cast(float, (int)gridDim.x);                                    \
    const unsigned long long smask = (unsigned long long)(unsigned)gridDim.x * 0x100000001ull;    \
    unsigned long long sm0 = 0, sm1 = 0;                                                         \
    (void)sx; (void)smask;
#define DEF_SKERNEL(NAME, BODY)                                                                 \
    __global__ __launch_bounds__(256) void NAME(const float* seed, float* sink, Out* out, int exec_mode) \
    {                                                                                           \
        KERNEL_PROLOGUE                                                                         \
        SGPR_PROLOGUE                                                                           \
        set_exec(exec_mode);                                                                    \
        for (int i = 0; i < ITER; ++i) { BODY; }                                                \
        a0 += (float)(sm0 + sm1);                                                               \
        KERNEL_EPILOGUE                                                                         \
    }
DEF_SKERNEL(calib_v_add_f32_sgpr_src, BODY_SGPR_SRC)
DEF_SKERNEL(calib_v_cndmask_sgpr_sel, BODY_CND_SGPR)
DEF_SKERNEL(calib_cmp_then_cndmask, BODY_CMP_CND)
DEF_SKERNEL(calib_v_cmp_to_sgpr, BODY_CMP_SGPR)

// v_cmp writing a scalar pair (each feeds nothing: pure issue cost)
__global__ __launch_bounds__(256) void calib_v_cmp_lt_f32(const float* seed, float* sink, Out* out, int exec_mode)
{
    KERNEL_PROLOGUE
    set_exec(exec_mode);
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("v_cmp_lt_f32 vcc, %0, %1\n\t v_cmp_lt_f32 vcc, %1, %2\n\t v_cmp_lt_f32 vcc, %2, %3\n\t"
                          "v_cmp_lt_f32 vcc, %3, %4\n\t v_cmp_lt_f32 vcc, %4, %5\n\t v_cmp_lt_f32 vcc, %5, %6\n\t"
                          "v_cmp_lt_f32 vcc, %6, %7\n\t v_cmp_lt_f32 vcc, %7, %0\n\t")
                     : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");
    }
    KERNEL_EPILOGUE
}

// packed f32: 8 accumulator PAIRS
#define PK_PROLOGUE                                                                              \
    v2f p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6},   \
        p6 = {a6, a7}, p7 = {a7, a0};                                                             \
    const v2f px = {x, y};
#define PK_EPILOGUE a0 = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;

#define BODY_PK3(INSN)                                                                          \
    asm volatile(REP8(INSN " %0, %8, %8, %0\n\t" INSN " %1, %8, %8, %1\n\t" INSN " %2, %8, %8, %2\n\t"  \
                      INSN " %3, %8, %8, %3\n\t" INSN " %4, %8, %8, %4\n\t" INSN " %5, %8, %8, %5\n\t"  \
                      INSN " %6, %8, %8, %6\n\t" INSN " %7, %8, %8, %7\n\t")                            \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)   \
                 : "v"(px))
#define BODY_PK2(INSN)                                                                          \
    asm volatile(REP8(INSN " %0, %8, %0\n\t" INSN " %1, %8, %1\n\t" INSN " %2, %8, %2\n\t"            \
                      INSN " %3, %8, %3\n\t" INSN " %4, %8, %4\n\t" INSN " %5, %8, %5\n\t"            \
                      INSN " %6, %8, %6\n\t" INSN " %7, %8, %7\n\t")                                  \
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)   \
                 : "v"(px))

#define DEF_PK_KERNEL(NAME, BODY)                                                               \
    __global__ __launch_bounds__(256) void NAME(const float* seed, float* sink, Out* out, int exec_mode) \
    {                                                                                           \
        KERNEL_PROLOGUE                                                                         \
        PK_PROLOGUE                                                                             \
        set_exec(exec_mode);                                                                    \
        for (int i = 0; i < ITER; ++i) { BODY; }                                                \
        asm volatile("s_mov_b64 exec, -1");                                                     \
        PK_EPILOGUE                                                                             \
        KERNEL_EPILOGUE                                                                         \
    }

DEF_PK_KERNEL(calib_v_pk_fma_f32, BODY_PK3("v_pk_fma_f32"))
DEF_PK_KERNEL(calib_v_pk_mul_f32, BODY_PK2("v_pk_mul_f32"))
DEF_PK_KERNEL(calib_v_pk_add_f32, BODY_PK2("v_pk_add_f32"))

// SALU alone: 64-bit mask arithmetic (the exec-mask juggling of step6)
__global__ __launch_bounds__(256) void calib_s_and_b64(const float* seed, float* sink, Out* out, int exec_mode)
{
    KERNEL_PROLOGUE
    unsigned long long m0 = (unsigned long long)(unsigned)gridDim.x * 0x100000001ull + 7ull, m1 = m0 + 1, m2 = m0 + 2, m3 = m0 + 3;
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("s_and_b64 %0, %0, %1\n\t s_or_b64 %1, %1, %2\n\t s_andn2_b64 %2, %2, %3\n\t s_xor_b64 %3, %3, %0\n\t"
                          "s_and_b64 %0, %0, %2\n\t s_or_b64 %1, %1, %3\n\t s_andn2_b64 %2, %2, %0\n\t s_xor_b64 %3, %3, %1\n\t")
                     : "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3) : : "scc");
    }
    a0 += (float)(m0 + m1 + m2 + m3);
    KERNEL_EPILOGUE
}

// VALU : SALU = 2 : 1 interleaved in every wave -- does the scalar unit issue beside the vector unit?
__global__ __launch_bounds__(256) void calib_mix_2fma_1salu(const float* seed, float* sink, Out* out, int exec_mode)
{
    KERNEL_PROLOGUE
    unsigned long long m0 = (unsigned long long)(unsigned)gridDim.x * 0x100000001ull + 7ull, m1 = m0 + 1, m2 = m0 + 2, m3 = m0 + 3;
    set_exec(exec_mode);
    for (int i = 0; i < ITER; ++i) {
        asm volatile(REP8("v_fma_f32 %0, %12, %13, %0\n\t v_fma_f32 %1, %12, %13, %1\n\t s_and_b64 %8, %8, %9\n\t"
                          "v_fma_f32 %2, %12, %13, %2\n\t v_fma_f32 %3, %12, %13, %3\n\t s_or_b64 %9, %9, %10\n\t"
                          "v_fma_f32 %4, %12, %13, %4\n\t v_fma_f32 %5, %12, %13, %5\n\t s_andn2_b64 %10, %10, %11\n\t"
                          "v_fma_f32 %6, %12, %13, %6\n\t v_fma_f32 %7, %12, %13, %7\n\t s_xor_b64 %11, %11, %8\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                       "+s"(m0), "+s"(m1), "+s"(m2), "+s"(m3)
                     : "v"(x), "v"(y) : "scc");
    }
    asm volatile("s_mov_b64 exec, -1");
    a0 += (float)(m0 + m1 + m2 + m3);
    KERNEL_EPILOGUE
}

// the packed exact division of one child box as step6 issues it (10 packed + 8 min/max per 6 quotients)
__global__ __launch_bounds__(256) void calib_slab_block(const float* seed, float* sink, Out* out, int exec_mode)
{
    KERNEL_PROLOGUE
    PK_PROLOGUE
    set_exec(exec_mode);
    for (int i = 0; i < ITER; ++i) {
        for (int k = 0; k < 3; ++k) {
            v2f tx, ty, tz;
            asm volatile("v_pk_add_f32 %[y], %[y], %[po] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                "v_pk_mul_f32 %[tx], %[x], %[px] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                "v_pk_mul_f32 %[tz], %[z], %[pz] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                "v_pk_mul_f32 %[ty], %[y], %[py] op_sel:[0,1] op_sel_hi:[1,1]\n\t"
                "v_pk_fma_f32 %[x], %[px], %[tx], %[x] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                "v_pk_fma_f32 %[z], %[pz], %[tz], %[z] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                "v_pk_fma_f32 %[y], %[py], %[ty], %[y] op_sel:[0,0,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
                "v_pk_fma_f32 %[x], %[x], %[px], %[tx] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                "v_pk_fma_f32 %[z], %[z], %[pz], %[tz] op_sel:[0,1,0] op_sel_hi:[1,1,1]\n\t"
                "v_pk_fma_f32 %[y], %[y], %[py], %[ty] op_sel:[0,1,0] op_sel_hi:[1,1,1]"
                : [x] "+v"(p0), [y] "+v"(p1), [z] "+v"(p2), [tx] "=&v"(tx), [ty] "=&v"(ty), [tz] "=&v"(tz)
                : [px] "v"(p3), [py] "v"(p4), [pz] "v"(p5), [po] "v"(px));
        }
    }
    asm volatile("s_mov_b64 exec, -1");
    PK_EPILOGUE
    KERNEL_EPILOGUE
}

typedef void (*kern_t)(const float*, float*, Out*, int);
struct Case { const char* name; kern_t k; int per_block; bool exec_sweep; };

int main(int argc, char** argv)
{
    const char* only = (argc > 1 && strcmp(argv[1], "all") != 0) ? argv[1] : nullptr;
    const int only_wpc = argc > 2 ? atoi(argv[2]) : 0;      // e.g. `valu_calib all 8` under the profiler
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# device %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
    float h_seed[16];
    for (int i = 0; i < 16; ++i) h_seed[i] = 1.0f + 0.001f * i;
    float *d_seed, *d_sink;
    Out* d_out;
    const int max_wgs = cus * 8;
    CK(hipMalloc(&d_seed, sizeof h_seed));
    CK(hipMalloc(&d_sink, (size_t)max_wgs * 256 * 4));
    CK(hipMalloc(&d_out, (size_t)max_wgs * 4 * sizeof(Out)));
    CK(hipMemcpy(d_seed, h_seed, sizeof h_seed, hipMemcpyHostToDevice));
    std::vector<Case> cases = {
        {"v_fma_f32", calib_v_fma_f32, 64, true}, {"v_mul_f32", calib_v_mul_f32, 64, false},
        {"v_add_f32", calib_v_add_f32, 64, false}, {"v_max_f32", calib_v_max_f32, 64, false},
        {"v_min3_f32", calib_v_min3_f32, 64, false}, {"v_rcp_f32", calib_v_rcp_f32, 64, true},
        {"v_mov_b32", calib_v_mov_b32, 64, false}, {"v_and_b32", calib_v_and_b32, 64, false},
        {"v_lshl_add_u32", calib_v_lshl_add_u32, 64, false}, {"v_cndmask_b32", calib_v_cndmask_b32, 64, false},
        {"v_cmp_lt_f32", calib_v_cmp_lt_f32, 64, false},
        {"v_sub_f32", calib_v_sub_f32, 64, false}, {"v_min_f32", calib_v_min_f32, 64, false},
        {"v_or_b32", calib_v_or_b32, 64, false}, {"v_xor_b32", calib_v_xor_b32, 64, false},
        {"v_lshlrev_b32", calib_v_lshlrev_b32, 64, false}, {"v_lshrrev_b32", calib_v_lshrrev_b32, 64, false},
        {"v_add_u32", calib_v_add_u32, 64, false}, {"v_sub_u32", calib_v_sub_u32, 64, false},
        {"v_mul_lo_u32", calib_v_mul_lo_u32, 64, false}, {"v_mul_u32_u24", calib_v_mul_u32_u24, 64, false},
        {"v_mad_u32_u24", calib_v_mad_u32_u24, 64, false}, {"v_bfe_u32", calib_v_bfe_u32, 64, false},
        {"v_bfi_b32", calib_v_bfi_b32, 64, false}, {"v_and_or_b32", calib_v_and_or_b32, 64, false},
        {"v_add3_u32", calib_v_add3_u32, 64, false}, {"v_med3_f32", calib_v_med3_f32, 64, false},
        {"v_mul_f32_e64", calib_v_mul_f32_e64, 64, false}, {"v_max_f32_e64", calib_v_max_f32_e64, 64, false},
        {"v_fmac_f32", calib_v_fmac_f32, 64, false}, {"v_mul_legacy_f32", calib_v_mul_legacy_f32, 64, false},
        {"v_add_f32(sgpr src)", calib_v_add_f32_sgpr_src, 64, false},
        {"v_cndmask(sgpr sel)", calib_v_cndmask_sgpr_sel, 64, false},
        {"cmp+cndmask pairs", calib_cmp_then_cndmask, 64, false},
        {"v_cmp -> sgpr pair", calib_v_cmp_to_sgpr, 64, false},
        {"v_pk_fma_f32", calib_v_pk_fma_f32, 64, true}, {"v_pk_mul_f32", calib_v_pk_mul_f32, 64, false},
        {"v_pk_add_f32", calib_v_pk_add_f32, 64, false},
        {"s_and_b64", calib_s_and_b64, 64, false}, {"mix_2fma_1salu(96)", calib_mix_2fma_1salu, 96, false},
        {"slab_block(30 pk)", calib_slab_block, 30, true},
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%-22s %5s %5s %12s %12s %10s %10s\n", "class", "w/SIMD", "exec", "cyc/inst/SIMD", "wall-based", "clock GHz", "wall us");
    for (const Case& c : cases) {
        if (only && !strstr(c.name, only)) continue;
        for (int wpc : {1, 2, 4, 8}) {
            if (only_wpc && wpc != only_wpc) continue;
            const int nexec = (c.exec_sweep && wpc == 8) ? 6 : 1;
            for (int em = 0; em < nexec; ++em) {
                const int grid = cus * wpc;
                for (int rep = 0; rep < 3; ++rep) {      // the last repetition is reported (warm clocks)
                    CK(hipEventRecord(e0, nullptr));
                    hipLaunchKernelGGL(c.k, dim3(grid), dim3(256), 0, nullptr, d_seed, d_sink, d_out, em);
                    CK(hipEventRecord(e1, nullptr));
                    CK(hipEventSynchronize(e1));
                }
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<Out> h((size_t)grid * 4);
                CK(hipMemcpy(h.data(), d_out, h.size() * sizeof(Out), hipMemcpyDeviceToHost));
                std::vector<double> cyc, clk;
                for (const Out& o : h) { cyc.push_back((double)o.cycles); clk.push_back((double)o.cycles / ((double)o.realtime * 10.0)); }
                std::sort(cyc.begin(), cyc.end());
                std::sort(clk.begin(), clk.end());
                const double med = cyc[cyc.size() / 2], ghz = clk[clk.size() / 2];   // s_memrealtime ticks at 100 MHz
                const double insts = (double)ITER * c.per_block;
                static const char* en[6] = {"all", "lo32", "hi32", "even", "lane0", "lo16"};
                printf("%-22s %5d %5s %12.3f %12.3f %10.3f %10.1f\n", c.name, wpc, en[em], med / (insts * wpc),
                       ms * 1e-3 * ghz * 1e9 / (insts * wpc), ghz, ms * 1e3);
            }
        }
    }
    return 0;
}
