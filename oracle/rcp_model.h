/* rcp_model.h -- bit-exact software model of gfx950's v_rcp_f32, driven by a table MEASURED on the GPU.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/).  The reference as shipped builds its kernels with
 * -cl-fast-relaxed-math (template/template.cpp:1192), which turns IntersectAABB's six divisions
 * (cl/extend.cl:31-35) into (b - o) * v_rcp_f32(d) and IntersectTri's f = 1 / a (extend.cl:17) into a bare
 * v_rcp_f32 on this GPU (read off the disassembly of oracle/_ref/ref_extend_fast.co).  v_rcp_f32 is a
 * hardware approximation (about 1 ulp), so a CPU restatement of THAT arithmetic needs the instruction's
 * actual result bits.  They are regular: for a normal input x = +-2^e * 1.m the result is
 * +-2^-e * rcp(1.m), so one table of the 2^23 significands -- read from the GPU by
 * oracle/rcp_probe.hip:refgpu_rcp_table at test time, never committed -- fixes the function; the rules for
 * zeros, infinities, NaNs, denormal inputs and results below the normal range are written out below.
 * oracle/rcp_probe.hip:refgpu_rcp_check compares this model with the instruction on ALL 2^32 inputs on the
 * GPU box (tests/test_gpu_shipped_flags.py demands zero mismatches), with this very header compiled for
 * the device.
 *
 * FP mode assumed: f32 denormals enabled, IEEE mode, DX10 clamp -- what the kernel descriptors of both the
 * reference's code objects and libuvrt_hip.so carry.
 */
#ifndef ORC_RCP_MODEL_H
#define ORC_RCP_MODEL_H

#include <stdint.h>

#ifdef __HIPCC__
#define ORC_RCP_FN __host__ __device__ static inline
#else
#define ORC_RCP_FN static inline
#endif

/* table[m] = bits of v_rcp_f32(1.m) for the 23-bit significand m, i.e. of a value in (0.5, 1] */
ORC_RCP_FN uint32_t orc_rcp_model_bits(uint32_t x, const uint32_t* table)
{
    const uint32_t sign = x & 0x80000000u;
    const uint32_t e = (x >> 23) & 0xFFu;
    const uint32_t m = x & 0x007FFFFFu;
    if (e == 255u) return m ? (x | 0x00400000u) : sign;           /* NaN -> quiet NaN (payload kept); inf -> 0 */
    /* the instruction flushes denormals on BOTH sides whatever the kernel's denormal mode says (measured on all 2^32
     * inputs, tests/tools/rcp_probe.py): a denormal input counts as zero, a result below 2^-126 comes out as zero */
    if (e == 0u) return sign | 0x7F800000u;                        /* +-0, +-denormal -> +-inf */
    const uint32_t r = table[m];
    const int32_t er = (int32_t)((r >> 23) & 0xFFu) - 127;        /* -1, or 0 for m == 0 */
    const int32_t E = er - ((int32_t)e - 127);                     /* unbiased exponent of the result: <= 126 */
    if (E < -126) return sign;
    return sign | ((uint32_t)(E + 127) << 23) | (r & 0x007FFFFFu);
}

#endif
