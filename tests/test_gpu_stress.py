"""GPU: maximum sizes and adversarial scenes.

* the reference's default workload at full size (positions/lange_route.xml: 2^25 photons over 12
  lamps, raytracer.h:31), one iteration, dose bit-identical to the oracle -- this size sits in
  the regime where generate.cl's float seed merges adjacent work-items;
* a synthetic 300 k-triangle soup through the native BVH builder (deposit replicas shrink, deeper
  tree);
* a scene built to force a deep traversal stack (> 16 entries: the global overflow stack) and one
  that overflows the reference's 32 entries (UVRT_ERR_STACK instead of the reference's silent
  memory corruption)."""
import numpy as np
import pytest

from conftest import GLB, ROUTE

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def host(pkg):
    from uvrt_amd import host
    return host


def test_reference_default_workload_full_size(host, orc, oscene, oroute):
    rt = host.RayTracer(GLB, ROUTE, device=0)
    assert rt.photonCount == 1 << 25 and len(rt.lamps()) == 12
    rt.ResetDosageMap()
    rt.ComputeDosageMap()
    rt.Shade()
    rt.Sync()
    comp = orc.Computation(oscene, oroute["lamps"], 1 << 25, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    comp.reset()
    comp.iteration()
    assert rt.photonsPerLight == comp.photonsPerLight == 2796202
    assert rt.ctx.seed == comp.SEED
    assert np.array_equal(rt.ctx.read_photon_map(0), comp.photonMap)
    assert np.array_equal(rt.ctx.read_photon_map(1), comp.maxPhotonMap)
    assert np.array_equal(bits(rt.read_dosage()), bits(comp.dose()))
    rt.close()


def soup(rng, T, extent, size):
    tris = np.zeros((T, 16), dtype=np.float32)
    c = rng.uniform(-extent, extent, size=(T, 1, 3))
    tris[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]] = (c + rng.normal(scale=size, size=(T, 3, 3))).reshape(T, 9).astype(np.float32)
    return tris


def test_large_synthetic_scene(host, orc):
    rng = np.random.default_rng(42)
    T = 300000
    tris = soup(rng, T, 4.0, 0.05)
    m = host.Mesh(tris=tris)
    ot = tris.copy()
    nodes, idx = orc.build_bvh(ot)
    assert m.nodesUsed == len(nodes) and np.array_equal(m.triIdx(), idx)
    assert np.array_equal(m.nodes().view(np.uint8), nodes.view(np.uint8))
    rt = host.RayTracer(mesh=m, device=0)
    rt.set_lamps([(0.1, -0.2, 30.0)])
    rt.photonCount = 400000
    rt.lightHeight = 0.5
    rt.ctx.set_record_hits(True)
    rt.ResetDosageMap()
    rt.ComputeDosageMap()
    rt.Sync()
    n = rt.photonsPerLight
    lp = (np.float32(0.1), np.float32(np.float32(m.floorHeight) + np.float32(0.5)), np.float32(-0.2))
    rays, _ = orc.generate(0, n, lp, rt.lightLength, 0)
    temp = np.zeros(T, dtype=np.int32)
    st = orc.extend(temp, ot, rays, nodes, idx)
    got = rt.ctx.read_rays(0, n)
    assert st["hits"] > 0.3 * n
    assert np.array_equal(bits(got["dist"]), bits(rays["dist"])) and np.array_equal(got["triID"], rays["triID"])
    assert np.array_equal(rt.ctx.read_photon_map(0), temp * 30.0)
    rt.close()
    m.close()


def chain_scene(depth, orc):
    """A hand-made BVH that keeps `depth` far children on every ray's traversal stack: a chain of
    inner nodes A_0 > A_1 > ... whose sibling at each level is a leaf B_k.  All boxes are cubes
    centred on the lamp, the A cubes larger than the B cubes, so for a ray starting inside them
    the A child always has the smaller (more negative) entry distance: descend into A_k, push B_k.
    (Boxes need not bound their triangles for the traversal to be well defined.)"""
    T = depth + 1
    tris = np.zeros((T, 16), dtype=np.float32)
    for k in range(T):                       # big triangles in front of the lamp at staggered depths
        zz = 3.0 + 0.37 * ((k * 7) % T)
        tris[k, 0:3] = (-4 + 0.1 * k, -4, zz); tris[k, 4:7] = (4, -4 + 0.05 * k, zz); tris[k, 8:11] = (0, 4, zz + 0.2)
    nodes = np.zeros(2 * depth + 2, dtype=orc.NODE_DT)

    def cube(n, h):
        n["minx"], n["miny"], n["minz"] = -h, -h, -h
        n["maxx"], n["maxy"], n["maxz"] = h, h, h

    parent = 0
    for k in range(depth):
        nodes[parent]["leftFirst"], nodes[parent]["triCount"] = 2 * k + 2, 0
        a, b = nodes[2 * k + 2], nodes[2 * k + 3]
        cube(a, 60.0 + depth - k)            # A_k: the rest of the chain (nearer entry)
        cube(b, 10.0 + k)                    # B_k: leaf with triangle k
        b["leftFirst"], b["triCount"] = k, 1
        parent = 2 * k + 2
    nodes[parent]["leftFirst"], nodes[parent]["triCount"] = depth, 1   # A_{depth-1} is a leaf
    cube(nodes[0], 100.0)
    return tris, nodes, np.arange(T, dtype=np.uint32)


@pytest.mark.parametrize("depth", [12, 20, 32])
def test_deep_stack_uses_overflow_and_matches_oracle(pkg, orc, depth):
    """Stack depths up to the reference's 32 entries (extend.cl:43): entries beyond the LDS part
    (16 in the default kernel, 8 in the staged one) live in the global overflow buffer."""
    tris, nodes, idx = chain_scene(depth, orc)
    n = 65536
    lp = (0.0, -0.5, 0.0)
    rays, _ = orc.generate(0, n, lp, 1.0, 0)
    temp = np.zeros(tris.shape[0], dtype=np.int32)
    st = orc.extend(temp, tris, rays, nodes, idx)
    assert st["max_stack"] == depth and st["hits"] > 0
    for variant in (0, 411, 501, 404, 500):
        c = pkg.capi.Ctx(0, dev=pkg.capi.needs_dev(variant))
        c.set_scene(tris, nodes, idx)
        c.resize_rays(n)
        c.set_record_hits(True)
        c.set_variant(variant)
        c.reset(False)
        c.seed = 0
        c.generate(lp, 1.0, 0, n)
        c.extend(n)
        c.sync()
        got = c.read_rays(0, n)
        assert np.array_equal(bits(got["dist"]), bits(rays["dist"])) and np.array_equal(got["triID"], rays["triID"])
        assert np.array_equal(c.read_counts(), temp)
        c.close()


def test_stack_overflow_is_reported_not_silent(pkg, orc):
    """33 pending far children: the reference writes past its 32-entry array (extend.cl:43,76);
    here every variant raises UVRT_ERR_STACK at the next sync."""
    tris, nodes, idx = chain_scene(33, orc)
    for variant in (0, 411, 501, 404, 500):
        c = pkg.capi.Ctx(0, dev=pkg.capi.needs_dev(variant))
        c.set_scene(tris, nodes, idx)
        c.set_variant(variant)
        c.resize_rays(4096)
        c.reset(False)
        c.generate((0.0, -0.5, 0.0), 1.0, 0, 4096)
        c.extend(4096)
        with pytest.raises(pkg.capi.UvrtError, match="32 stack entries"):
            c.sync()
        c.sync()                             # the flag is cleared once reported
        c.close()


def test_malformed_bvh_is_rejected(pkg, orc):
    tris = np.zeros((2, 16), dtype=np.float32)
    idx = np.array([0, 1], dtype=np.uint32)
    c = pkg.capi.Ctx(0)
    bad = np.zeros(4, dtype=orc.NODE_DT)
    bad[0]["leftFirst"], bad[0]["triCount"] = 2, 0
    bad[2]["leftFirst"], bad[2]["triCount"] = 0, 0        # cycle: child points back to the root pair
    bad[3]["leftFirst"], bad[3]["triCount"] = 0, 1
    with pytest.raises(pkg.capi.UvrtError):
        c.set_scene(tris, bad, idx)
    bad2 = np.zeros(1, dtype=orc.NODE_DT)
    bad2[0]["leftFirst"], bad2[0]["triCount"] = 1, 5       # leaf beyond triIdx
    with pytest.raises(pkg.capi.UvrtError):
        c.set_scene(tris, bad2, idx)
    bad3 = np.zeros(1, dtype=orc.NODE_DT)
    bad3[0]["leftFirst"], bad3[0]["triCount"] = 7, 0       # children out of range
    with pytest.raises(pkg.capi.UvrtError):
        c.set_scene(tris, bad3, idx)
    with pytest.raises(pkg.capi.UvrtError):
        c.set_scene(tris, np.zeros(1, dtype=orc.NODE_DT), np.array([0, 9], dtype=np.uint32))
    c.close()


def test_reference_maximum_photon_count(host, orc, oscene, oroute):
    """maxPhotonCount = 2^26 (raytracer.h:30) in ONE launch: beyond the reference's own int overflow
    of `32 * photonCount` (raytracer.cpp:137) and deep into the regime where the
    float seed merges neighbouring work-items.  Counts and dose equal the oracle."""
    rt = host.RayTracer(GLB, ROUTE, device=0)
    rt.set_lamps(rt.lamps()[5:6])
    rt.photonCount = 1 << 26
    rt.ResetDosageMap()
    rt.ComputeDosageMap()
    rt.Shade()
    rt.Sync()
    comp = orc.Computation(oscene, oroute["lamps"][5:6], 1 << 26, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    comp.reset()
    comp.iteration()
    assert rt.photonsPerLight == comp.photonsPerLight == 1 << 26
    assert np.array_equal(rt.ctx.read_photon_map(0), comp.photonMap)
    assert np.array_equal(bits(rt.read_dosage()), bits(comp.dose()))
    rt.close()


def test_second_init_restarts_seed_and_state(host, orc, oscene, oroute):
    """Model reload calls Init again (userinterface.cpp:239-240): generate.cl is rebuilt, so SEED
    restarts at 0; the mirror creates a fresh context."""
    import ctypes
    rt = host.RayTracer(GLB, ROUTE, device=0)
    rt.set_lamps(rt.lamps()[:1])
    rt.photonCount = 50000
    rt.ResetDosageMap()
    rt.ComputeDosageMap()
    rt.Sync()
    first = rt.ctx.read_photon_map(0)
    assert rt.ctx.seed != 0
    rt._L.uvrt_host_rt_init(rt._h, rt.mesh._h)                      # RayTracer::Init(mesh) again
    rt.ctx = host._BorrowedCtx(rt._L.uvrt_host_rt_ctx(rt._h), rt.mesh.triangleCount)
    assert rt.ctx.seed == 0
    rt.set_lamps(rt.lamps()[:1])
    rt.photonCount = 50000
    rt.ResetDosageMap()
    rt.ComputeDosageMap()
    rt.Sync()
    assert np.array_equal(rt.ctx.read_photon_map(0), first)
    rt.close()
