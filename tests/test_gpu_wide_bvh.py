"""GPU: the OPT-IN 4-wide BVH walk (include/uvrt.h uvrt_set_wide_bvh) against the default walk
(the reference's visit order) and the oracle.  Closest hit is order-independent except where two accepted hits
tie exactly in t (extend.cl:25: strict <, first found wins) or an earlier, almost equally distant hit culls
the box of a later one (extend.cl:66-76): `dist` bits must be identical on EVERY ray, `triID` on every ray
without such a tie; the rays that differ are counted and reported, and the per-triangle counts must be equal
whenever there are none."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def lamp_pos(orc, oscene, oroute, k):
    comp = orc.Computation(oscene, oroute["lamps"], 1 << 16, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    return tuple(float(x) for x in comp.lamp_world_pos(oroute["lamps"][k]))


@pytest.mark.parametrize("flavour", [0, 1])
def test_wide_walk_equals_the_reference_order_walk(pkg, orc, oscene, oroute, flavour):
    n = 1920 * 1080
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        c.resize_rays(n)
        c.set_record_hits(True)
        c.set_flavour(flavour)
        total_diff = 0
        for li, seed in ((0, 0), (4, 77), (8, 0x5151), (11, 9)):
            lp = lamp_pos(orc, oscene, oroute, li)
            res = []
            for wide in (False, True):
                c.set_wide_bvh(wide)
                c.reset(False)
                c.seed = seed
                c.generate(lp, oroute["lightLength"], 0, n)
                c.extend(n)
                c.sync()
                res.append((c.read_rays(0, n), c.read_counts()))
            (r2, cnt2), (r4, cnt4) = res
            assert np.array_equal(bits(r4["dist"]), bits(r2["dist"])), "lamp %d" % li
            diff = np.flatnonzero(r4["triID"] != r2["triID"])
            total_diff += diff.size
            print("lamp %2d flavour %d: %d of %d rays end on another triangle at the same distance" % (li, flavour, diff.size, n))
            # such rays hit (dist finite) and the two triangles are hit at bit-equal t
            assert (r2["dist"][diff] != np.float32(1e30)).all()
            if diff.size == 0:
                assert np.array_equal(cnt4, cnt2)
            else:
                assert np.abs(cnt4.astype(np.int64) - cnt2).sum() == 2 * diff.size
            assert cnt4.sum() == cnt2.sum() and cnt2.sum() > 0.8 * n
        assert total_diff <= 8          # the test room has no coincident geometry: expect 0
    finally:
        c.close()


def test_wide_walk_on_degenerate_and_small_scenes(pkg, orc):
    """Two-triangle scene with a leaf root (CalibratePower's, raytracer.cpp:166-187), a 3-triangle scene
    (a node with a leaf child and an inner child) and a 70-triangle soup, against the oracle."""
    rng = np.random.default_rng(3)
    for T in (2, 3, 70):
        tris = np.zeros((T, 16), dtype=np.float32)
        ctr = rng.uniform(-1, 1, (T, 3)).astype(np.float32) * np.float32(0.8)
        for k in range(3):
            tris[:, 4 * k:4 * k + 3] = ctr + rng.uniform(-0.5, 0.5, (T, 3)).astype(np.float32)
        nodes, idx = orc.build_bvh(tris)
        n = 50000
        lp = (0.05, -0.9, 0.1)
        rays, _ = orc.generate(0, n, lp, 1.0, 3)
        temp = np.zeros(T, dtype=np.int32)
        orc.extend(temp, tris, rays, nodes, idx)
        c = pkg.capi.Ctx(0)
        try:
            c.set_scene(tris, nodes, idx)
            c.resize_rays(n)
            c.set_record_hits(True)
            c.set_wide_bvh(True)
            c.reset(False)
            c.seed = 3
            c.generate(lp, 1.0, 0, n)
            c.extend(n)
            c.sync()
            got = c.read_rays(0, n)
            assert np.array_equal(bits(got["dist"]), bits(rays["dist"])), T
            assert np.array_equal(got["triID"], rays["triID"]), T
            assert np.array_equal(c.read_counts(), temp) and temp.sum() > 0
        finally:
            c.close()


def test_wide_walk_full_computation_dose(pkg, orc, oscene, oroute):
    """Pipelined host loop with the wide walk: 2 lamps x 3 iterations, dose bits against the oracle (holds as
    long as no ray of the run is order-dependent, which the first test establishes for this scene)."""
    from uvrt_amd import host
    from conftest import GLB, ROUTE
    rt = host.RayTracer(GLB, ROUTE, device=0)
    try:
        rt.set_lamps(rt.lamps()[:2])
        rt.photonCount = 600000
        rt.ctx.set_wide_bvh(True)
        rt.ResetDosageMap()
        for _ in range(3):
            rt.ComputeDosageMap()
            rt.Shade()
        dose = rt.read_dosage()
    finally:
        rt.close()
    comp = orc.Computation(oscene, oroute["lamps"][:2], 600000, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    comp.reset()
    for _ in range(3):
        comp.iteration()
    assert np.array_equal(bits(dose), bits(comp.dose())) and dose.any()
