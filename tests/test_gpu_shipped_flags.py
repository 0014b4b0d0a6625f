"""GPU: the reference AS SHIPPED.  The reference builds its kernels with -cl-fast-relaxed-math -cl-mad-enable
-cl-single-precision-constant (template/template.cpp:1192); oracle/Makefile compiles cl/extend.cl and cl/generate.cl
with exactly those options for gfx950 (oracle/_ref/ref_*_fast.co).  uvrt_set_flavour(ctx, 2) reproduces what that
extend kernel computes -- t = (b - o) * v_rcp_f32(d) in the slab test, f = v_rcp_f32(a) in the triangle test -- and must
equal it bit for bit: (dist bits, triID) per ray and the count vector.  The CPU oracle follows in flavour 2 through a
software model of v_rcp_f32 (oracle/rcp_model.h) whose table is read from this GPU and which is checked here against the
instruction on all 2^32 inputs.  Default flavour 0 stays the canonical strict arithmetic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def ref(orc):
    L = orc.refgpu()
    if L is None or not orc.refgpu_have_shipped():
        pytest.skip("oracle/_ref/ref_extend_fast.co not built (needs /root/reference at build time)")
    return L


@pytest.fixture(scope="module")
def table(ref, orc):
    t = orc.refgpu_rcp_table()
    orc.set_rcp_table(t)
    return t


def lamp(orc, oscene, oroute, k):
    comp = orc.Computation(oscene, oroute["lamps"], 1 << 16, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    return tuple(float(x) for x in comp.lamp_world_pos(oroute["lamps"][k]))


def test_rcp_model_equals_the_instruction_on_all_inputs(orc, table):
    """oracle/rcp_model.h built from the 2^23-entry table against v_rcp_f32 on every binary32 bit pattern"""
    assert table[0] == 0x3F800000                                  # rcp(1) = 1
    exact = (np.float32(1) / (np.arange(1 << 23, dtype=np.uint32) | 0x3F800000).view(np.float32)).view(np.uint32)
    d = table.astype(np.int64) - exact.astype(np.int64)
    print("v_rcp_f32 vs RN(1/x) over the 2^23 significands: equal %.4f, -1 ulp %.4f, +1 ulp %.4f, beyond %d"
          % ((d == 0).mean(), (d == -1).mean(), (d == 1).mean(), int((np.abs(d) > 1).sum())))
    assert np.abs(d).max() <= 1                                    # the 1-ulp approximation the ISA documents
    bad, first = orc.refgpu_rcp_check(table)
    assert bad == 0, "model != v_rcp_f32 on %d inputs, e.g. (x, hw, model) %s" % (bad, [tuple(hex(int(v)) for v in r) for r in first[:8]])


@pytest.mark.parametrize("li,seed", [(0, 0), (5, 0x79044923), (11, 0x1234567)])
def test_flavour2_equals_the_reference_kernel_built_with_its_own_flags(ref, table, pkg, orc, oscene, oroute, li, seed):
    n = 4096 * 256
    lp = lamp(orc, oscene, oroute, li)
    rays, _ = orc.generate(0, n, lp, oroute["lightLength"], seed)
    r_rays = rays.copy()
    r_counts, ms = orc.refgpu_extend(r_rays, oscene.tris, oscene.nodes, oscene.triIdx, shipped=True)
    s_rays = rays.copy()
    s_counts, ms_strict = orc.refgpu_extend(s_rays, oscene.tris, oscene.nodes, oscene.triIdx)
    # the product, flavour 2, on the same rays (SEED semantics are generate's business: canonical here)
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        c.resize_rays(n)
        c.set_record_hits(True)
        c.set_flavour(2)
        c.reset(False)
        c.seed = seed
        c.generate(lp, oroute["lightLength"], 0, n)
        c.extend(n)
        c.sync()
        got = c.read_rays(0, n)
        counts = c.read_counts()
    finally:
        c.close()
    for f in ("dirx", "diry", "dirz", "origy"):
        assert np.array_equal(bits(got[f]), bits(rays[f]))
    assert np.array_equal(bits(got["dist"]), bits(r_rays["dist"])), int((bits(got["dist"]) != bits(r_rays["dist"])).sum())
    assert np.array_equal(got["triID"], r_rays["triID"])
    assert np.array_equal(counts, r_counts) and counts.sum() > 0.8 * n
    # the oracle in flavour 2 (v_rcp_f32 through the measured table)
    o_rays = rays.copy()
    o_counts = np.zeros(oscene.T, dtype=np.int32)
    orc.set_flavour(2)
    try:
        orc.extend(o_counts, oscene.tris, o_rays, oscene.nodes, oscene.triIdx)
    finally:
        orc.set_flavour(0)
    assert np.array_equal(bits(o_rays["dist"]), bits(r_rays["dist"])) and np.array_equal(o_rays["triID"], r_rays["triID"])
    assert np.array_equal(o_counts, r_counts)
    # how far the shipped arithmetic is from the strict build of the same source
    diff_tri = int((r_rays["triID"] != s_rays["triID"]).sum())
    diff_bits = int((bits(r_rays["dist"]) != bits(s_rays["dist"])).sum())
    print("lamp %d: reference extend.cl with its own flags %.3f ms vs strict %.3f ms per %d rays; against the strict build "
          "%d rays hit another triangle, %d rays differ in dist bits" % (li, ms, ms_strict, n, diff_tri, diff_bits))
    assert diff_tri <= 64


def test_flavour2_whole_computation_through_every_tracing_path(ref, table, pkg, orc, oscene, oroute):
    """2 lamps x 2 iterations through the per-launch calls (pipelined) and the batched path: counts, f64 maps and dose
    equal the oracle's in flavour 2; against flavour 0 the dose differs on a handful of triangles only."""
    n = 300000
    length = oroute["lightLength"]
    lamps = [lamp(orc, oscene, oroute, 0), lamp(orc, oscene, oroute, 7)] * 2
    durations = [60.0, 30.0, 60.0, 30.0]
    T = oscene.T

    def oracle(flavour):
        orc.set_flavour(flavour)
        try:
            pm, mm, temp = np.zeros(T), np.zeros(T), np.zeros(T, dtype=np.int32)
            seed = 0
            for k, lp in enumerate(lamps):
                rays, seed = orc.generate(0, n, lp, length, seed)
                orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
                orc.accumulate(pm, mm, temp, durations[k])
            return pm, mm, orc.compute_dosage(pm, oscene.tris, 2 * n, np.float32(44.0197))
        finally:
            orc.set_flavour(0)

    pm2, mm2, dose2 = oracle(2)
    _, _, dose0 = oracle(0)
    for how in ("calls", "batched"):
        c = pkg.capi.Ctx(0)
        try:
            c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
            c.set_flavour(2)
            c.resize_rays(n)
            c.reset(True)
            c.seed = 0
            if how == "batched":
                c.trace_batch(lamps, length, 0, n)
                ops = np.zeros(4, dtype=pkg.capi.REPLAY_OP_DT)
                for k in range(4):
                    ops[k] = (durations[k], 1 if k == 3 else 0, 0, 2 * n, 44.0197, 1.0, 0)
                c.replay_batch(ops)
            else:
                for k, lp in enumerate(lamps):
                    c.generate(lp, length, 0, n)
                    c.extend(n)
                    c.accumulate(durations[k])
                c.compute_dosage(0, 2 * n, 44.0197)
            c.sync()
            assert np.array_equal(c.read_photon_map(0), pm2) and np.array_equal(c.read_photon_map(1), mm2), how
            assert np.array_equal(bits(c.read_dosage()), bits(dose2)), how
        finally:
            c.close()
    rel = np.abs(dose2.astype(np.float64) - dose0) / np.maximum(np.abs(dose0), 1e-30)
    far = int((rel > 1e-4).sum())
    print("flavour 2 vs flavour 0 dose: %d of %d triangles beyond 1e-4 (a photon on a neighbouring triangle)" % (far, T))
    assert far <= 40


def test_flavour_range_and_the_wide_walk_in_flavour2(ref, table, pkg, orc, oscene, oroute):
    """uvrt_set_flavour accepts 0, 1, 2; the opt-in 4-wide walk (uvrt_set_wide_bvh) carries the shipped-flags arithmetic too:
    its counts equal the oracle's in flavour 2 on this scene (the wide walk's own caveat -- order-dependent exact ties -- stands)."""
    n = 500000
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        with pytest.raises(pkg.capi.UvrtError, match="0, 1 or 2"):
            c.set_flavour(3)
        c.set_flavour(2)
        c.set_wide_bvh(True)
        c.resize_rays(n)
        lp = lamp(orc, oscene, oroute, 3)
        c.reset(False)
        c.seed = 11
        c.generate(lp, oroute["lightLength"], 0, n)
        c.extend(n)
        c.sync()
        got = c.read_counts()
    finally:
        c.close()
    rays, _ = orc.generate(0, n, lp, oroute["lightLength"], 11)
    temp = np.zeros(oscene.T, dtype=np.int32)
    orc.set_flavour(2)
    try:
        orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
    finally:
        orc.set_flavour(0)
    assert np.array_equal(got, temp)


def test_flavour2_on_hand_made_rays_equals_the_reference_kernel(ref, table, pkg, orc, oscene):
    """Rays random photons reach once per ten million: exact zeros in the direction (v_rcp_f32 gives inf, (b - o) * inf is +-inf
    or NaN where the origin lies on the slab plane and the hardware's min / max drop the NaN), origins on box planes, |d| > 1,
    subnormal components (v_rcp_f32 flushes them: inf).  Flavour 2 has ONE form for all of them -- the instructions the
    reference's own-flags build executes -- so product, oracle model and that kernel must still agree bit for bit."""
    from test_gpu_adversarial import make_rays
    nodes = oscene.nodes
    inner = nodes[nodes["triCount"] == 0]
    ox, oz = float(inner["minx"][3]), float(inner["maxz"][7])
    rng = np.random.default_rng(3)
    planes_y = np.concatenate([inner["miny"][:200], inner["maxy"][:200]])
    dirs, oys = [], []
    axes = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    for k in range(256 * 24):
        kind = k % 8
        if kind == 0:
            d = axes[rng.integers(6)]
        elif kind in (1, 2):
            a = rng.normal(size=3); a[rng.integers(3)] = 0.0 if kind == 1 else -0.0; d = a / np.linalg.norm(a)
        elif kind == 3:
            a = rng.normal(size=3); d = 3.5 * a / np.linalg.norm(a)                      # not unit length
        elif kind == 4:
            a = rng.normal(size=3); d = a / np.linalg.norm(a); d[rng.integers(3)] = 1e-41   # subnormal component
        elif kind == 5:
            a = rng.normal(size=3); d = a / np.linalg.norm(a); d[rng.integers(3)] = -1e-30
        else:
            a = rng.normal(size=3); d = a / np.linalg.norm(a)
        dirs.append(d)
        oys.append(planes_y[rng.integers(planes_y.size)] if k % 2 else rng.uniform(-1.4, 1.3))
    rays = make_rays(dirs, (ox, oz), oys)
    r_rays = rays.copy()
    r_counts, _ = orc.refgpu_extend(r_rays, oscene.tris, oscene.nodes, oscene.triIdx, shipped=True)
    o_rays = rays.copy()
    o_counts = np.zeros(oscene.T, dtype=np.int32)
    orc.set_flavour(2)
    try:
        orc.extend(o_counts, oscene.tris, o_rays, oscene.nodes, oscene.triIdx)
    finally:
        orc.set_flavour(0)
    assert np.array_equal(bits(o_rays["dist"]), bits(r_rays["dist"])) and np.array_equal(o_rays["triID"], r_rays["triID"])
    assert np.array_equal(o_counts, r_counts) and r_counts.sum() > 0.3 * rays.size
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        c.resize_rays(rays.size)
        c.set_record_hits(True)
        c.set_flavour(2)
        c.reset(False)
        c.write_rays(rays)
        c.extend(rays.size)
        c.sync()
        got = c.read_rays(0, rays.size)
        assert np.array_equal(bits(got["dist"]), bits(r_rays["dist"])), int((bits(got["dist"]) != bits(r_rays["dist"])).sum())
        assert np.array_equal(got["triID"], r_rays["triID"])
        assert np.array_equal(c.read_counts(), r_counts)
    finally:
        c.close()


def test_reference_generate_built_with_its_own_flags(ref, orc, oscene, oroute):
    """generate.cl with the reference's own flags on this GPU: reported against the strict build of the same source
    (fast-math may re-associate the f32 seed sum of generate.cl:13).  Nothing of the product depends on it:
    the product's generate is pinned to the strict build (test_gpu_reference_kernels.py)."""
    n = 2048 * 256
    lp = lamp(orc, oscene, oroute, 0)
    orc.refgpu_reload()
    strict, ms_s = orc.refgpu_generate(n, lp, oroute["lightLength"])
    orc.refgpu_reload()
    shipped, ms_f = orc.refgpu_generate(n, lp, oroute["lightLength"], shipped=True)
    same = np.ones(n, dtype=bool)
    for f in ("dirx", "diry", "dirz", "origx", "origy", "origz"):
        same &= bits(strict[f]) == bits(shipped[f])
    print("reference generate.cl with its own flags: %.3f ms (strict build %.3f ms) per %d work-items; %.4f of the rays equal "
          "the strict build's bit for bit" % (ms_f, ms_s, n, same.mean()))
    d = np.sqrt(shipped["dirx"].astype(np.float64) ** 2 + shipped["diry"].astype(np.float64) ** 2 + shipped["dirz"].astype(np.float64) ** 2)
    assert np.abs(d - 1.0).max() < 1e-5
    assert np.all(shipped["origx"] == np.float32(lp[0])) and np.all(shipped["origz"] == np.float32(lp[2]))
