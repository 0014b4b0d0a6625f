"""GPU: the per-lamp hot-record set-up (csrc/uvrt_hotset.hip: visit statistics -> selection of the most visited
node-pair records -> renumbering).  Traversal results never depend on the renumbering, so the parity tests cannot
see a poor selection; these tests read the renumbering back (uvrt_read_record_perm) and compare it with the visit
counts of the ORACLE's traversal of the same sample rays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEEP = 175      # records the traversal kernel serves from LDS (uvrt_device.h TOP6_MAX)
SAMPLE = 32768  # photons of the launch whose visits are counted (uvrt_ctx.h hot_sample)


def pair_order(nodes):
    """node index of every pair record: the inner nodes breadth-first from node 0, as uvrt_set_scene numbers them"""
    order = [0]
    qi = 0
    while qi < len(order):
        l = int(nodes["leftFirst"][order[qi]])
        for ch in (l, l + 1):
            if nodes["triCount"][ch] == 0:
                order.append(ch)
        qi += 1
    return np.array(order, dtype=np.int64)


def lamp_pos(orc, oscene, oroute, k):
    comp = orc.Computation(oscene, oroute["lamps"], 1 << 16, oroute["lightHeight"], oroute["lightLength"], oroute["lightIntensity"])
    return tuple(float(x) for x in comp.lamp_world_pos(oroute["lamps"][k]))


@pytest.mark.parametrize("lamp,seed", [(0, 0), (5, 0x79044923), (11, 7)])
def test_hot_records_are_the_most_visited_ones(pkg, orc, oscene, oroute, lamp, seed):
    order = pair_order(oscene.nodes)
    P = order.size
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        n = 65536
        c.resize_rays(n)
        lp = lamp_pos(orc, oscene, oroute, lamp)
        c.seed = seed
        c.generate(lp, oroute["lightLength"], 0, n)
        perm = c.read_record_perm(P)
        # a permutation; hot and cold records each keep their index order
        assert np.array_equal(np.sort(perm), np.arange(P, dtype=np.uint32))
        hot = np.flatnonzero(perm < KEEP)
        cold = np.flatnonzero(perm >= KEEP)
        assert hot.size == KEEP and np.all(np.diff(perm[hot].astype(np.int64)) > 0) and np.all(np.diff(perm[cold].astype(np.int64)) > 0)
        # the oracle's visit counts of the same sample rays (exact arithmetic; the device counts in fast arithmetic)
        rays, _ = orc.generate(0, SAMPLE, lp, oroute["lightLength"], seed)
        visits = orc.extend_visit_hist(oscene.tris, rays, oscene.nodes, oscene.triIdx)[order].astype(np.int64)
        best = np.sort(visits)[::-1][:KEEP].sum()
        got = visits[hot].sum()
        assert got >= 0.998 * best, (got, best)
        assert visits[hot].min() >= 0.9 * np.sort(visits)[::-1][KEEP - 1]
        assert perm[0] < KEEP                                  # the root is visited by every ray
        # a second launch from the same lamp reuses the cached renumbering; another lamp gets its own
        c.generate(lp, oroute["lightLength"], 0, n)
        assert np.array_equal(c.read_record_perm(P), perm)
        c.generate(lamp_pos(orc, oscene, oroute, (lamp + 3) % 12), oroute["lightLength"], 0, n)
        assert not np.array_equal(c.read_record_perm(P), perm)
        c.set_hot_records(0)
        c.generate(lp, oroute["lightLength"], 0, n)
        assert np.array_equal(c.read_record_perm(P), np.arange(P, dtype=np.uint32))
    finally:
        c.close()


def test_selection_with_and_without_the_tree_walk(pkg, orc, oscene, oroute, monkeypatch):
    """k_select_hot looks for the hot records among the first 8192 records and walks the tree from the root only when a
    record beyond them belongs to the set (frontier check).  Forced through the developer knob UVRT_HOT_DIRECT (read in
    uvrt_create): no direct candidates at all, 64 of them (the check must fire: the hot records reach beyond 64), and
    the default -- one and the same renumbering."""
    P = pair_order(oscene.nodes).size
    lp = lamp_pos(orc, oscene, oroute, 4)
    perms = []
    for direct in (None, "0", "64", "1000"):
        if direct is None:
            monkeypatch.delenv("UVRT_HOT_DIRECT", raising=False)
        else:
            monkeypatch.setenv("UVRT_HOT_DIRECT", direct)
        c = pkg.capi.Ctx(0)
        try:
            c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
            c.resize_rays(65536)
            c.seed = 99
            c.generate(lp, oroute["lightLength"], 0, 65536)
            perms.append(c.read_record_perm(P))
        finally:
            c.close()
    assert np.flatnonzero(perms[0] < KEEP).size == KEEP
    for q in perms[1:]:
        assert np.array_equal(q, perms[0])


def test_more_lamp_positions_than_cache_entries(pkg, orc, oscene, oroute):
    """70 lamp positions through the 64-entry cache of renumberings (least recently used entries are rebuilt): counts
    of the last launches still equal the oracle's."""
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        n = 20000
        c.resize_rays(n)
        base = lamp_pos(orc, oscene, oroute, 2)
        for k in range(70):
            lp = (base[0] + 0.01 * k, base[1], base[2] - 0.005 * k)
            c.reset(False)
            c.seed = k
            c.generate(lp, oroute["lightLength"], 0, n)
            c.extend(n)
            if k in (0, 63, 64, 69):
                rays, _ = orc.generate(0, n, lp, oroute["lightLength"], k)
                temp = np.zeros(oscene.T, dtype=np.int32)
                orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
                assert np.array_equal(c.read_counts(), temp), k
        c.sync()
    finally:
        c.close()


def test_tiny_scenes_keep_the_breadth_first_order(pkg, orc):
    """Fewer than 129 inner nodes: everything fits in the LDS cache, no statistics are taken."""
    rng = np.random.default_rng(3)
    T = 100
    tris = np.zeros((T, 16), dtype=np.float32)
    ctr = rng.uniform(-1, 1, (T, 3)).astype(np.float32)
    for k in range(3):
        tris[:, 4 * k:4 * k + 3] = ctr + rng.uniform(-0.2, 0.2, (T, 3)).astype(np.float32)
    nodes, idx = orc.build_bvh(tris)
    P = pair_order(nodes).size
    assert P <= 128
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(tris, nodes, idx)
        n = 32768
        c.resize_rays(n)
        c.generate((0.0, 0.0, 0.0), 1.0, 0, n)
        assert np.array_equal(c.read_record_perm(P), np.arange(P, dtype=np.uint32))
        c.extend(n)
        rays, _ = orc.generate(0, n, (0.0, 0.0, 0.0), 1.0, 0)
        temp = np.zeros(T, dtype=np.int32)
        orc.extend(temp, tris, rays, nodes, idx)
        assert np.array_equal(c.read_counts(), temp)
    finally:
        c.close()


def test_hot_records_on_a_triangle_soup(pkg, orc):
    """Another tree shape (50 000 random triangles in the room's box, deeper and bushier than the room): the selection
    still covers what the true top 175 cover, and counts with the renumbering equal the oracle's."""
    import bench
    tris = bench.soup_triangles(50000, seed=5)
    nodes, idx = orc.build_bvh(tris)
    order = pair_order(nodes)
    P = order.size
    lp = (0.1, -0.4, 1.0)
    n = 65536
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(tris, nodes, idx)
        c.resize_rays(n)
        c.reset(False)
        c.seed = 3
        c.generate(lp, 1.0, 0, n)
        perm = c.read_record_perm(P)
        assert np.array_equal(np.sort(perm), np.arange(P, dtype=np.uint32))
        hot = np.flatnonzero(perm < KEEP)
        rays, _ = orc.generate(0, n, lp, 1.0, 3)
        visits = orc.extend_visit_hist(tris, rays[:SAMPLE].copy(), nodes, idx)[order].astype(np.int64)
        best = np.sort(visits)[::-1][:KEEP].sum()
        # (a soup's rays take more steps than the room's: the 64-step cut of the statistics costs a little here -- 99.3 %)
        assert visits[hot].sum() >= 0.98 * best, (visits[hot].sum(), best)
        c.extend(n)
        temp = np.zeros(tris.shape[0], dtype=np.int32)
        orc.extend(temp, tris, rays, nodes, idx)
        assert np.array_equal(c.read_counts(), temp)
    finally:
        c.close()
