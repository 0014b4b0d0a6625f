"""CPU: the arithmetic identity behind extend v2's slab test (csrc/uvrt_kernels.hip slab<>()):

    RN32(a / d)  ==  RN32( f64(a) * RN64(1 / f64(d)) )      for binary32 a, d with |d| <= 1
                                                            and a quotient that is zero or normal

numpy's float32 division and float64 multiply are IEEE, so this checks the identity itself on
random operands, on operands built to put a/d next to a rounding midpoint, and on the special
values the kernel lets through (zeros, infinities from huge quotients)."""
import numpy as np


def both(a, d):
    a = a.astype(np.float32)
    d = d.astype(np.float32)
    with np.errstate(all="ignore"):
        exact = a / d
        fast = (a.astype(np.float64) * (1.0 / d.astype(np.float64))).astype(np.float32)
    return exact, fast


def assert_same(a, d):
    exact, fast = both(a, d)
    same = (exact.view(np.uint32) == fast.view(np.uint32)) | (np.isnan(exact) & np.isnan(fast))
    bad = np.flatnonzero(~same)
    assert bad.size == 0, (a[bad[:5]], d[bad[:5]], exact[bad[:5]], fast[bad[:5]])


def test_random_operands():
    rng = np.random.default_rng(7)
    for _ in range(20):
        n = 2_000_000
        d = rng.uniform(-1, 1, n).astype(np.float32)
        d[d == 0] = 1.0
        a = (rng.normal(size=n) * 10.0 ** rng.uniform(-6, 3, n)).astype(np.float32)
        assert_same(a, d)


def test_random_bit_patterns():
    rng = np.random.default_rng(11)
    n = 4_000_000
    # any finite float a with |a| >= 2^-100 or a == 0; any d with 2^-126 <= |d| <= 1
    a = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32).view(np.float32)
    a = np.where(np.isfinite(a) & ((np.abs(a) >= 2.0 ** -100) | (a == 0)), a, np.float32(1.5))
    e = rng.integers(1, 128, n).astype(np.uint32)                # biased exponent 1..127 -> |d| < 2
    m = rng.integers(0, 2 ** 23, n).astype(np.uint32)
    s = rng.integers(0, 2, n).astype(np.uint32)
    d = ((s << 31) | (e << 23) | m).view(np.float32)
    d = np.where(np.abs(d) <= 1, d, np.float32(0.75))
    assert_same(a, d)


def test_quotients_next_to_midpoints():
    """a = RN32(m * d) for m a midpoint between two adjacent floats: a/d is as close to a
    rounding boundary as the operands allow."""
    rng = np.random.default_rng(13)
    n = 3_000_000
    d = rng.uniform(2.0 ** -20, 1, n).astype(np.float32) * rng.choice([-1, 1], n).astype(np.float32)
    q = rng.uniform(1, 2, n).astype(np.float32) * (2.0 ** rng.integers(-20, 20, n)).astype(np.float32)
    mid = q.astype(np.float64) + 0.5 * np.spacing(q).astype(np.float64)        # exact midpoints
    for k in (-1, 0, 1):
        a = np.nextafter((mid * d.astype(np.float64)).astype(np.float32), np.float32(np.inf if k > 0 else -np.inf)) \
            if k else (mid * d.astype(np.float64)).astype(np.float32)
        assert_same(a, d)


def test_special_values():
    a = np.array([0.0, -0.0, 1e30, -1e30, 3e38, 1.0, 1e-30, 2.0 ** -100, 5.0], dtype=np.float32)
    for dv in (1.0, -1.0, 2.0 ** -126, -(2.0 ** -126), 2.0 ** -149, 1e-20, 0.3, -0.7):
        assert_same(a, np.full(a.shape, dv, dtype=np.float32))


# ------------------------------------------------------------------------------------------------
# extend v5/v6: the packed f32 form  q0 = a*y; r = fma(-d, q0, a); q = fma(r, y, q0),  y = RN32(1/d)
# (csrc/uvrt_extend6.hip).  The proof is the exhaustive GPU run (tests/tools/div3_exhaustive.hip,
# profiles/r01/r01_div3_exhaustive.log); this is the same check on the CPU (glibc fmaf is exact) over
# random, extreme-divisor and next-to-midpoint operands.
_C_SRC = r"""
#include <math.h>
#include <stdint.h>
#include <string.h>
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint64_t rng(uint64_t* s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
long check(long n, uint64_t seed)
{
    long bad = 0;
    uint64_t s = seed;
    for (long i = 0; i < n; i++) {
        uint64_t x = rng(&s), y = rng(&s);
        int mode = (int)(i & 7);
        uint32_t md = (uint32_t)(y & 0x7fffff);
        if (mode == 1) md |= 0x7ffff0;                 /* divisor significand next to all ones */
        if (mode == 2) md &= 0xf;                      /* ... next to a power of two */
        int ed = 127 - (int)((y >> 23) % 40);
        float d = u2f(((uint32_t)((y >> 63) & 1) << 31) | ((uint32_t)ed << 23) | md);
        if (fabsf(d) > 1.0f) d = 1.0f;
        float a;
        if (mode >= 3) {                               /* quotient next to a rounding midpoint */
            uint32_t mq = (uint32_t)(x & 0x7fffff);
            int eq = 127 - 20 + (int)((x >> 23) % 40);
            float q = u2f(((uint32_t)eq << 23) | mq);
            double mid = (double)q + 0.5 * (double)(u2f(f2u(q) + 1) - q);
            a = (float)(mid * (double)d);
            a = u2f(f2u(a) + (uint32_t)((int)((x >> 40) % 5) - 2));
        } else {
            uint32_t ma = (uint32_t)(x & 0x7fffff);
            int ea = 127 - 30 + (int)((x >> 23) % 50);
            a = u2f(((uint32_t)((x >> 63) & 1) << 31) | ((uint32_t)ea << 23) | ma);
        }
        float ex = a / d;
        if (!isfinite(ex) || fabsf(ex) < 1.2e-38f) continue;
        float yy = (float)(1.0 / (double)d);
        float q0 = a * yy, r = fmaf(-d, q0, a), q = fmaf(r, yy, q0);
        if (f2u(q) != f2u(ex)) bad++;
    }
    return bad;
}
"""


def test_packed_three_instruction_division_equals_ieee_quotient(tmp_path):
    import ctypes
    import subprocess
    src = tmp_path / "div3.c"
    src.write_text(_C_SRC)
    lib = tmp_path / "libdiv3.so"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(lib), str(src), "-lm"])
    L = ctypes.CDLL(str(lib))
    L.check.restype = ctypes.c_long
    L.check.argtypes = [ctypes.c_long, ctypes.c_uint64]
    assert L.check(30_000_000, 88172645463325252) == 0


def test_proof_condition_range_tests_on_bit_patterns():
    """uvrt_extend6.hip outside_proof_conditions(): the refill decides with unsigned range tests on the bit patterns
    (|x| in [lo, hi] <=> bits(|x|) - bits(lo) <= bits(hi) - bits(lo)) which rays leave the packed exact division;
    this is the float-comparison form the proof conditions are stated in, restated both ways in numpy."""
    import itertools
    rng = np.random.default_rng(7)

    def by_floats(r):
        ay, a = np.abs(r[:, 3]), np.abs(r[:, :3])
        with np.errstate(invalid="ignore"):
            return ((~(a >= np.float32(8.6736174e-19))).any(1) | (~(a <= np.float32(1.0))).any(1) |
                    ((ay != 0) & (ay < np.float32(7.888609e-31))) | (~(ay <= np.float32(1e9))))

    def by_bits(r):
        u = r.view(np.uint32) & np.uint32(0x7FFFFFFF)
        lo, one = np.uint32(0x21800000), np.uint32(0x3F800000)
        worst = (u[:, :3] - lo).max(1)                                   # wraps below lo
        uo, ylo, yhi = u[:, 3], np.uint32(0x0D800000), np.uint32(0x4E6E6B28)
        return (worst > one - lo) | ((uo != 0) & ((uo - ylo) > (yhi - ylo)))

    assert np.float32(8.6736174e-19).view(np.uint32) == 0x21800000 and np.float32(7.888609e-31).view(np.uint32) == 0x0D800000
    assert np.float32(1e9).view(np.uint32) == 0x4E6E6B28
    x = rng.integers(0, 2 ** 32, size=(1 << 20, 4), dtype=np.uint64).astype(np.uint32).view(np.float32)
    f32 = np.float32
    edge = np.array([0.0, -0.0, 8.6736174e-19, np.nextafter(f32(8.6736174e-19), f32(0)), 1.0, np.nextafter(f32(1), f32(2)),
                     7.888609e-31, np.nextafter(f32(7.888609e-31), f32(0)), 1e9, np.nextafter(f32(1e9), f32(2e9)),
                     np.inf, np.nan, 1e-45, 0.5, -0.5, -1.0], dtype=np.float32)
    e = np.array(list(itertools.product(edge, repeat=4)), dtype=np.float32)
    with np.errstate(over="ignore"):
        for arr in (x, e):
            assert np.array_equal(by_floats(arr), by_bits(arr))
