"""GPU: hand-made rays and degenerate scenes fed straight into extend (uvrt_write_rays), against
the oracle bit for bit.  These force the paths that random photons reach about once per ten
million rays: zero direction components (0/0 = NaN in the slab test, where OpenCL's select-form
min/max and the hardware's IEEE min/max disagree), |d| > 1 and tiny origins (outside the proof
conditions of the reciprocal division), and leaves with 15+ triangles."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def run_both(pkg, orc, tris, nodes, idx, rays, variants=(0, 401, 501, 801, 405, 500)):
    o_rays = rays.copy()
    o_rays["dist"] = np.float32(1e30)
    o_rays["triID"] = 0
    temp = np.zeros(tris.shape[0], dtype=np.int32)
    st = orc.extend(temp, tris, o_rays, nodes, idx)
    for dev in (False, True):       # the product library, then the developer build's extra kernel variants
        vs = [v for v in variants if pkg.capi.needs_dev(v) == dev]
        if not vs:
            continue
        c = pkg.capi.Ctx(0, dev=dev)
        c.set_scene(tris, nodes, idx)
        c.resize_rays(rays.size)
        c.set_record_hits(True)
        for v in vs:
            c.set_variant(v)
            c.reset(False)
            c.write_rays(rays)
            c.extend(rays.size)
            c.sync()
            got = c.read_rays(0, rays.size)
            assert np.array_equal(bits(got["dist"]), bits(o_rays["dist"])), "variant %d" % v
            assert np.array_equal(got["triID"], o_rays["triID"]), "variant %d" % v
            assert np.array_equal(c.read_counts(), temp), "variant %d" % v
        c.close()
    return st, o_rays


def make_rays(dirs, origin, oys):
    r = np.zeros(len(dirs), dtype=np.dtype([("dirx", "<f4"), ("diry", "<f4"), ("dirz", "<f4"), ("origx", "<f4"),
                                             ("origy", "<f4"), ("origz", "<f4"), ("dist", "<f4"), ("triID", "<u4")]))
    d = np.asarray(dirs, dtype=np.float32)
    r["dirx"], r["diry"], r["dirz"] = d[:, 0], d[:, 1], d[:, 2]
    r["origx"], r["origz"] = np.float32(origin[0]), np.float32(origin[1])
    r["origy"] = np.asarray(oys, dtype=np.float32)
    r["dist"] = np.float32(1e30)
    return r


def test_zero_direction_components_and_origins_on_box_planes(pkg, orc, oscene):
    """Origin x/z taken from a BVH node plane and origin y from node planes, directions with exact
    zeros: (bound - o) / d becomes 0/0 for some slabs.  The select-form min/max must be used."""
    nodes = oscene.nodes
    inner = nodes[nodes["triCount"] == 0]
    ox = float(inner["minx"][3])          # exactly a slab plane
    oz = float(inner["maxz"][7])
    rng = np.random.default_rng(3)
    dirs, oys = [], []
    axes = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    planes_y = np.concatenate([inner["miny"][:200], inner["maxy"][:200]])
    for k in range(6000):
        kind = k % 6
        if kind == 0:
            d = axes[rng.integers(6)]                                   # two zero components
        elif kind == 1:
            a = rng.normal(size=3); a[rng.integers(3)] = 0.0; d = a / np.linalg.norm(a)   # one zero
        elif kind == 2:
            a = rng.normal(size=3); a[rng.integers(3)] = -0.0; d = a / np.linalg.norm(a)  # negative zero
        elif kind == 3:
            a = rng.normal(size=3); d = a / np.linalg.norm(a)
        elif kind == 4:
            a = rng.normal(size=3); a[1] = 0.0; d = a / np.linalg.norm(a)  # horizontal
        else:
            d = (0.0, rng.choice([-1.0, 1.0]), 0.0)                      # straight up / down
        dirs.append(d)
        oys.append(planes_y[rng.integers(planes_y.size)] if k % 2 else rng.uniform(-1.4, 1.3))
    rays = make_rays(dirs, (ox, oz), oys)
    n = (rays.size // 64) * 64
    st, o = run_both(pkg, orc, oscene.tris, oscene.nodes, oscene.triIdx, rays[:n])
    assert st["hits"] > 0.3 * n


def test_outside_the_fast_path_conditions(pkg, orc, oscene):
    """|d| > 1, subnormal / tiny direction components and tiny non-zero origins: lanes take the
    IEEE-division path; results still equal the oracle."""
    rng = np.random.default_rng(4)
    n = 4096
    a = rng.normal(size=(n, 3))
    d = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    d[0::7] *= np.float32(3.5)                     # not unit length
    d[1::7, 0] = np.float32(1e-41)                 # subnormal component
    d[2::7, 2] = np.float32(-3e-39)
    d[3::7, 1] = np.float32(1e-30)
    oys = rng.uniform(-1.4, 1.3, n).astype(np.float32)
    oys[4::7] = np.float32(2.0 ** -120)            # tiny non-zero origin component
    oys[5::7] = np.float32(0.0)
    for origin in ((0.0, 0.0), (2.0 ** -110, -(2.0 ** -105)), (-0.255, -3.31)):
        rays = make_rays(d, origin, oys)
        run_both(pkg, orc, oscene.tris, oscene.nodes, oscene.triIdx, rays)


def test_big_leaf_scene(pkg, orc):
    """All centroids coincide: the builder cannot split, the root is ONE leaf with 40 triangles
    (leaf count code 15 + side table), and a second scene whose BVH has 20-triangle leaves."""
    rng = np.random.default_rng(9)
    T = 40
    tris = np.zeros((T, 16), dtype=np.float32)
    for k in range(T):    # nested coplanar triangles whose vertex sums are exactly (0, 0, 3z): one centroid
        a_k, b_k, z = np.float32(0.2 + 0.05 * k), np.float32(0.1 + 0.03 * k), np.float32(2.0)
        tris[k, 0:3] = (-a_k, -b_k, z); tris[k, 4:7] = (a_k, -b_k, z); tris[k, 8:11] = (0.0, 2 * b_k, z)
    ot = tris.copy()
    nodes, idx = orc.build_bvh(ot)
    a = rng.normal(size=(8192, 3)); a[:, 2] = np.abs(a[:, 2]) + 1.0
    d = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    rays = make_rays(d, (0.01, 0.0), rng.uniform(-0.3, 0.3, 8192))
    st, _ = run_both(pkg, orc, ot, nodes, idx, rays)
    assert st["hits"] > 1000
    assert nodes[0]["triCount"] == T and len(nodes) == 1       # un-splittable: the root is the only node
    # hand-made BVH: root -> two leaves of 20 triangles each
    nodes2 = np.zeros(4, dtype=orc.NODE_DT)
    nodes2[0]["leftFirst"], nodes2[0]["triCount"] = 2, 0
    for j, (lo, hi) in enumerate(((0, 20), (20, 40))):
        nd = nodes2[2 + j]
        v = ot[lo:hi][:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(-1, 3)
        nd["minx"], nd["miny"], nd["minz"] = v.min(0)
        nd["maxx"], nd["maxy"], nd["maxz"] = v.max(0)
        nd["leftFirst"], nd["triCount"] = lo, hi - lo
    run_both(pkg, orc, ot, nodes2, np.arange(T, dtype=np.uint32), rays)


def test_degenerate_triangles_give_nan_dose_like_the_reference(pkg, orc, oscene):
    """Zero-area triangles: computeDosage divides by area * photonsPerLight = 0 (shade.cl:36-39,
    SURVEY App. B): with no photon on them that is 0/0 = NaN.  Same NaN positions as the oracle,
    identical bits everywhere else; the degenerate triangles do not disturb the trace either."""
    tris = oscene.tris[:3000].copy()
    tris[10, 4:7] = tris[10, 0:3]; tris[10, 8:11] = tris[10, 0:3]              # a point
    tris[11, 0:3] = (0.5, 0.25, 1.0); tris[11, 4:7] = (1.5, 0.25, 1.0); tris[11, 8:11] = (2.5, 0.25, 1.0)   # collinear
    ot = tris.copy()
    nodes, idx = orc.build_bvh(ot)
    n = 65536
    lp = (-0.255, -0.995, -3.31)
    rays, _ = orc.generate(0, n, lp, 1.0, 0)
    temp = np.zeros(3000, dtype=np.int32)
    orc.extend(temp, ot, rays, nodes, idx)
    pm, mm = np.zeros(3000), np.zeros(3000)
    counts = temp.copy()
    orc.accumulate(pm, mm, temp, 60.0)
    ref = orc.compute_dosage(pm, ot, n, 44.0)
    c = pkg.capi.Ctx(0)
    c.set_scene(ot, nodes, idx)
    c.resize_rays(n)
    c.reset(True)
    c.generate(lp, 1.0, 0, n)
    c.extend(n)
    c.sync()
    assert np.array_equal(c.read_counts(), counts) and counts[10] == 0 and counts[11] == 0
    c.accumulate(60.0)
    c.compute_dosage(0, n, 44.0)
    c.sync()
    got = c.read_dosage()
    c.close()
    assert np.isnan(ref[10]) and np.isnan(ref[11])
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(bits(got[ok]), bits(ref[ok])) and counts.sum() > 0
