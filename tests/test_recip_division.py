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
