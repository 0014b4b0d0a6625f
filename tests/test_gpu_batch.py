"""GPU: batched tracing (include/uvrt.h uvrt_trace_batch / uvrt_replay_batch / uvrt_reduce_batch*) against the
per-launch sequence of the reference's host loop (raytracer.cpp:66-88, myapp.cpp:156-163) and the oracle:
per-launch counts, f64 maps, dose, colours and the SEED chain must be bit-identical; ray-range sharding with
ONE reduction per batch must equal the unsharded computation."""
import numpy as np
import pytest

from conftest import GLB, ROUTE

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def bits64(a):
    return np.ascontiguousarray(a).view(np.uint64)


def lamp_pos(orc, oscene, oroute, k):
    comp = orc.Computation(oscene, oroute["lamps"], 1 << 16, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    return tuple(float(x) for x in comp.lamp_world_pos(oroute["lamps"][k]))


def make_ops(pkg, durations, shade_at, ppl):
    ops = np.zeros(len(durations), dtype=pkg.capi.REPLAY_OP_DT)
    for k, d in enumerate(durations):
        ops[k] = (d, 1 if k in shade_at else 0, shade_at.get(k, 0), ppl * (k + 1), 44.0197, 100.0, k & 1)
    return ops


@pytest.mark.parametrize("n", [100001, 65536])
def test_batch_equals_the_per_launch_sequence_and_the_oracle(pkg, orc, oscene, oroute, n):
    """Five launches from three lamps (two lamp columns repeat, so launches are regrouped internally),
    a ray count that is not a multiple of 64; replay with a Shade after launches 2 (max map) and 4 (sum)."""
    length = oroute["lightLength"]
    order = [0, 5, 0, 9, 5]
    lamps = [lamp_pos(orc, oscene, oroute, k) for k in order]
    durations = [60.0, 30.0, 45.0, 10.0, 25.0]
    shade_at = {2: 1, 4: 0}
    ops = make_ops(pkg, durations, shade_at, n)
    a = pkg.capi.Ctx(0)
    b = pkg.capi.Ctx(0)
    try:
        for c in (a, b):
            c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
            c.resize_rays(n)
            c.reset(True)
            c.seed = 0x1234
        # per-launch sequence
        per_launch = []
        snap = {}
        for k, lp in enumerate(lamps):
            a.generate(lp, length, 0, n)
            a.extend(n)
            per_launch.append(a.read_counts())
            a.accumulate(durations[k])
            if k in shade_at:
                a.shade(int(ops[k]["which_map"]), int(ops[k]["photons_per_light"]), float(ops[k]["scaled_power"]),
                        float(ops[k]["min_value"]), int(ops[k]["threshold_view"]))
                snap[k] = (a.read_dosage(), a.read_color())
        # batch
        b.trace_batch(lamps, length, 0, n)
        assert b.seed == a.seed
        for k in range(len(lamps)):
            assert np.array_equal(b.read_batch_counts(k), per_launch[k]), k
        b.replay_batch(ops)
        assert np.array_equal(bits64(b.read_photon_map(0)), bits64(a.read_photon_map(0)))
        assert np.array_equal(bits64(b.read_photon_map(1)), bits64(a.read_photon_map(1)))
        assert np.array_equal(bits(b.read_dosage()), bits(snap[4][0])) and np.array_equal(bits(b.read_color()), bits(snap[4][1]))
        # the oracle on launch 3
        rays, _ = orc.generate(0, n, lamps[3], length, pkg.capi.seed_next(lamps[2], length, pkg.capi.seed_next(
            lamps[1], length, pkg.capi.seed_next(lamps[0], length, 0x1234))))
        temp = np.zeros(oscene.T, dtype=np.int32)
        orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
        assert np.array_equal(temp, per_launch[3]) and temp.sum() > 0.5 * n
        # a second batch on the same context, replayed WITHOUT the fold (the un-folded replica path)
        b.trace_batch(lamps[:2], length, 7, 1000)
        b.replay_batch(make_ops(pkg, durations[:2], {1: 0}, 1000))
        a.generate(lamps[0], length, 7, 1000); a.extend(1000); a.accumulate(durations[0])
        a.generate(lamps[1], length, 7, 1000); a.extend(1000); a.accumulate(durations[1])
        assert np.array_equal(bits64(b.read_photon_map(0)), bits64(a.read_photon_map(0)))
        assert b.seed == a.seed
    finally:
        a.close()
        b.close()


def test_a_new_record_renumbering_at_the_same_address_is_not_mistaken_for_the_old_one(pkg, orc, oscene, oroute):
    """uvrt_set_record_perm(P1), batch, uvrt_set_record_perm(P2) with the same n -- the context's buffer keeps
    its address -- then a batch from the SAME lamp column: the per-launch records must be re-laid for P2 (the kernel
    starts at P2[root]); 70 lamps through the 64-entry hot cache make an entry be recycled under a key that matches too."""
    n = 40000
    length = oroute["lightLength"]
    lp = lamp_pos(orc, oscene, oroute, 0)
    rng = np.random.default_rng(5)
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        nodes = oscene.nodes                                     # inner nodes reachable from the root
        q, i = [0], 0
        while i < len(q):
            l = int(nodes[q[i]]["leftFirst"]); i += 1
            q += [l + k for k in (0, 1) if nodes[l + k]["triCount"] == 0]
        npairs = len(q)
        expect = []
        seed = 0
        for _ in range(3):
            rays, seed2 = orc.generate(0, n, lp, length, seed)
            temp = np.zeros(oscene.T, dtype=np.int32)
            orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
            expect.append(temp)
            seed = seed2
        c.seed = 0
        for k, perm in enumerate([rng.permutation(npairs), rng.permutation(npairs), None]):
            c.set_record_perm(None if perm is None else perm.astype(np.uint32))
            c.trace_batch([lp], length, 0, n)
            got = c.read_batch_counts(0)
            c.replay_batch(make_ops(pkg, [1.0], {}, n))
            assert np.array_equal(got, expect[k]), "batch %d traced against stale records" % k
        # hot entries: 70 distinct lamps through the 64-entry cache, then the first lamp again
        c.seed = 0
        for j in range(70):
            q = (lp[0] + 0.001 * (j + 1), lp[1], lp[2])
            c.trace_batch([q], length, 0, 20000)
            c.replay_batch(make_ops(pkg, [1.0], {}, 20000))
        c.seed = 0
        c.trace_batch([lp], length, 0, n)
        assert np.array_equal(c.read_batch_counts(0), expect[0])
        c.replay_batch(make_ops(pkg, [1.0], {}, n))
    finally:
        c.close()


def test_batch_call_order_errors_and_reset(pkg, orc, oscene, oroute):
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        lp = lamp_pos(orc, oscene, oroute, 0)
        with pytest.raises(pkg.capi.UvrtError, match="no traced batch"):
            c.replay_batch(make_ops(pkg, [1.0], {}, 10))
        c.trace_batch([lp], 1.0, 0, 5000)
        with pytest.raises(pkg.capi.UvrtError, match="not been replayed"):
            c.trace_batch([lp], 1.0, 0, 5000)
        with pytest.raises(pkg.capi.UvrtError, match="2 operations"):
            c.replay_batch(make_ops(pkg, [1.0, 2.0], {}, 10))
        c.reset(True)                      # drops the traced batch
        c.trace_batch([lp], 1.0, 0, 5000)
        cnt = c.read_batch_counts(0)
        c.replay_batch(make_ops(pkg, [2.0], {}, 5000))
        assert np.array_equal(c.read_photon_map(0), cnt.astype(np.float64) * 2.0) and cnt.sum() > 2000
        with pytest.raises(pkg.capi.UvrtError, match=r"\[1,64\]"):
            c.trace_batch([lp] * 65, 1.0, 0, 100)
        with pytest.raises(pkg.capi.UvrtError, match="no communicator"):
            c.trace_batch([lp], 1.0, 0, 100)
            c.reduce_batch()
    finally:
        c.close()


def _host_rt(pkg, photons, lamps_n, view):
    from uvrt_amd import host
    rt = host.RayTracer(GLB, ROUTE, device=0)
    rt.set_lamps(rt.lamps()[:lamps_n])
    rt.photonCount = photons
    rt.viewMode = view
    rt.thresholdView = view == host.VIEW_MAXPOWER
    return rt


@pytest.mark.parametrize("lamps_n,iterations,photons,view", [(3, 3, 180000, 0), (12, 6, 120000, 1), (1, 8, 921600, 0)])
def test_batched_raytracer_equals_the_host_loop(pkg, lamps_n, iterations, photons, view):
    """RayTracer::ComputeIterationsBatched against ComputeDosageMap(); Shade(); per iteration (myapp.cpp:
    156-163): 12 lamps x 6 iterations = 72 launches spans two batches; 1 lamp x 8 x 921 600 is the shape
    of BASELINE configs[1] doubled."""
    a = _host_rt(pkg, photons, lamps_n, view)
    b = _host_rt(pkg, photons, lamps_n, view)
    try:
        a.maxIterations = b.maxIterations = iterations
        a.ResetDosageMap()
        for _ in range(iterations):
            a.ComputeDosageMap()
            a.Shade()
            a.currIterations = a.currIterations + 1
        b.ResetDosageMap()
        b.ComputeIterationsBatched(iterations)
        assert b.photonMapSize == a.photonMapSize and b.currIterations == a.currIterations == iterations
        assert np.array_equal(bits(b.read_dosage()), bits(a.read_dosage()))
        assert np.array_equal(bits(b.ctx.read_color()), bits(a.ctx.read_color()))
        for w in (0, 1):
            assert np.array_equal(bits64(b.ctx.read_photon_map(w)), bits64(a.ctx.read_photon_map(w)))
        assert a.ctx.seed == b.ctx.seed and a.read_dosage().any()
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("world", [2, 3])
def test_ray_range_shards_with_one_reduction_per_batch_equal_the_whole(pkg, world):
    """BASELINE configs[3] on one device: `world` instances each trace their global-id range of every
    launch into private planes, ONE sum of the planes per batch (uvrt_reduce_batch_group: contexts of one
    device), then every instance replays accumulate + Shade -- all must hold the unsharded dose bits."""
    from uvrt_amd import host
    lamps_n, iterations, photons = 2, 4, 200002
    one = _host_rt(pkg, photons, lamps_n, 0)
    shards = [_host_rt(pkg, photons, lamps_n, 0) for _ in range(world)]
    try:
        one.ResetDosageMap()
        one.ComputeIterationsBatched(iterations)
        want = one.read_dosage()
        for r, rt in enumerate(shards):
            rt.ResetDosageMap()
            rt.SetRayRange(r, world)
        host.compute_iterations_batched_group(shards, iterations)
        for rt in shards:
            assert np.array_equal(bits(rt.read_dosage()), bits(want))
            assert np.array_equal(bits64(rt.ctx.read_photon_map(1)), bits64(one.ctx.read_photon_map(1)))
            assert rt.ctx.seed == one.ctx.seed
        assert want.any()
    finally:
        one.close()
        for rt in shards:
            rt.close()


def test_rccl_all_reduce_of_the_planes_single_rank_communicator(pkg, orc, oscene, oroute):
    """The native collective (librccl opened by the library, ncclCommInitRank + ncclAllReduce(int32) on the
    context's stream) with the world the box allows: one rank.  The planes must come back unchanged and the
    replay must equal the un-reduced one; librccl must really be mapped into the process."""
    lamps = [lamp_pos(orc, oscene, oroute, k) for k in (0, 1)]
    n = 150000
    a = pkg.capi.Ctx(0)
    b = pkg.capi.Ctx(0)
    try:
        for c in (a, b):
            c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
            c.reset(True)
        b.comm_init_rank(pkg.capi.comm_unique_id(), 0, 1)
        assert "librccl" in open("/proc/self/maps").read()
        ops = make_ops(pkg, [60.0, 20.0], {1: 0}, n)
        a.trace_batch(lamps, 1.0, 0, n)
        want = [a.read_batch_counts(k) for k in range(2)]
        a.replay_batch(ops)
        b.trace_batch(lamps, 1.0, 0, n)
        b.reduce_batch()
        for k in range(2):
            assert np.array_equal(b.read_batch_counts(k), want[k])
        b.replay_batch(ops)
        assert np.array_equal(bits(a.read_dosage()), bits(b.read_dosage())) and a.read_dosage().any()
        b.comm_destroy()
    finally:
        a.close()
        b.close()


def test_baseline_config_1_shape_host_loop(pkg, orc, oscene, oroute):
    """BASELINE configs[1]: "1280x720 4-bounce on one MI355X" = 921 600 photons x 4 waves from lamp 0 through the
    reference's host loop (launch pipelining on, hot-record cache on: the defaults), dose bits vs the oracle."""
    rt = _host_rt(pkg, 921600, 1, 0)
    try:
        rt.ResetDosageMap()
        for _ in range(4):
            rt.ComputeDosageMap()
            rt.Shade()
            rt.currIterations = rt.currIterations + 1
        dose = rt.read_dosage()
        assert rt.photonsPerLight == 921600 and rt.photonMapSize == 4 * 921600
    finally:
        rt.close()
    comp = orc.Computation(oscene, oroute["lamps"][:1], 921600, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    comp.reset()
    for _ in range(4):
        comp.iteration()
    assert np.array_equal(bits(dose), bits(comp.dose())) and (dose != 0).sum() > 20000


def test_batch_of_64_launches_many_lamp_columns_and_chunks(pkg, orc, oscene, oroute, monkeypatch):
    """The largest batch (64 launches) over all 12 lamps of lange_route.xml in a scrambled order (12 lamp columns,
    5-6 launches each), with the chunk size forced down so that every column is traced in several chunks; the
    per-launch counts of a sample of launches against the oracle, the maps against the per-launch replay on the CPU."""
    monkeypatch.setenv("UVRT_BATCH_CHUNK_MB", "1")           # 1 MB = 2 planes of 20 011 rays (n_pad 20 032 x 16 B)
    n = 20011
    rng = np.random.default_rng(11)
    order = [int(x) for x in rng.integers(0, 12, 64)]
    lamps = [lamp_pos(orc, oscene, oroute, k) for k in order]
    durations = [float(d) for d in rng.integers(1, 90, 64)]
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        c.reset(True)
        c.seed = 5
        c.trace_batch(lamps, oroute["lightLength"], 3, n)
        planes = [c.read_batch_counts(k) for k in range(64)]
        c.replay_batch(make_ops(pkg, durations, {63: 0}, n))
        pm, mm = c.read_photon_map(0), c.read_photon_map(1)
    finally:
        c.close()
    seed = 5
    want_pm, want_mm = np.zeros(oscene.T), np.zeros(oscene.T)
    for k in range(64):
        if k in (0, 17, 40, 63):
            rays, nxt = orc.generate(3, n, lamps[k], oroute["lightLength"], seed)
            temp = np.zeros(oscene.T, dtype=np.int32)
            orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
            assert np.array_equal(temp, planes[k]), k
        else:
            nxt = pkg.capi.seed_next(lamps[k], oroute["lightLength"], seed)
        seed = nxt
        cnt = planes[k].copy()
        orc.accumulate(want_pm, want_mm, cnt, durations[k])
    assert np.array_equal(bits64(pm), bits64(want_pm)) and np.array_equal(bits64(mm), bits64(want_mm)) and pm.any()


def test_batches_and_single_launches_interleave_on_one_context(pkg, orc, oscene, oroute):
    """The same twelve launches (4 lamps x 3 iterations) as (a) single launches, (b) batch / single launches / batch
    mixed on one context, with read-backs in between: maps, dose, colours and SEED identical at the end, and the
    read-backs in the middle equal the single-launch sequence at the same point."""
    length = oroute["lightLength"]
    n = 50003
    lamps4 = [lamp_pos(orc, oscene, oroute, k) for k in (2, 6, 7, 10)]
    seq = [(lamps4[k % 4], 10.0 + k) for k in range(12)]
    a = pkg.capi.Ctx(0)
    b = pkg.capi.Ctx(0)
    try:
        for c in (a, b):
            c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
            c.resize_rays(n)
            c.reset(True)
        mid_a = {}
        for k, (lp, d) in enumerate(seq):
            a.generate(lp, length, 0, n)
            a.extend(n)
            a.accumulate(d)
            if k in (4, 7):
                mid_a[k] = a.read_photon_map(0)
        a.shade(0, 3 * n, 44.0, 100.0, 0)

        def ops_for(part, shade_last=False):
            ops = np.zeros(len(part), dtype=pkg.capi.REPLAY_OP_DT)
            for j, (_, d) in enumerate(part):
                ops[j] = (d, 0, 0, 1, 1.0, 1.0, 0)
            if shade_last:
                ops[-1] = (part[-1][1], 1, 0, 3 * n, 44.0, 100.0, 0)
            return ops

        b.trace_batch([lp for lp, _ in seq[:5]], length, 0, n)
        b.replay_batch(ops_for(seq[:5]))
        assert np.array_equal(bits64(b.read_photon_map(0)), bits64(mid_a[4]))
        for lp, d in seq[5:8]:                      # three single launches (pipelined lanes) in between
            b.generate(lp, length, 0, n)
            b.extend(n)
            b.accumulate(d)
        assert np.array_equal(bits64(b.read_photon_map(0)), bits64(mid_a[7]))
        b.trace_batch([lp for lp, _ in seq[8:]], length, 0, n)
        b.replay_batch(ops_for(seq[8:], shade_last=True))
        assert a.seed == b.seed
        for w in (0, 1):
            assert np.array_equal(bits64(b.read_photon_map(w)), bits64(a.read_photon_map(w)))
        assert np.array_equal(bits(b.read_dosage()), bits(a.read_dosage())) and a.read_dosage().any()
        assert np.array_equal(bits(b.read_color()), bits(a.read_color()))
    finally:
        a.close()
        b.close()


def test_group_collective_rejects_what_it_cannot_reduce(pkg, orc, oscene, oroute):
    """uvrt_comm_init_all / uvrt_reduce_batch_group argument checks on real contexts: RCCL wants one rank per device
    (two contexts of one device are refused -- uvrt_reduce_batch_group sums those without it), a group needs batches
    of one shape, a context takes one communicator, and uvrt_reduce_batch needs a communicator and a batch."""
    lamps = [lamp_pos(orc, oscene, oroute, k) for k in (0, 1, 2)]
    n = 20000
    a = pkg.capi.Ctx(0)
    b = pkg.capi.Ctx(0)
    try:
        for c in (a, b):
            c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
            c.reset(True)
        with pytest.raises(pkg.capi.UvrtError, match="share device"):
            pkg.capi.comm_init_all([a, b])
        with pytest.raises(pkg.capi.UvrtError, match="no traced batch"):
            a.reduce_batch()
        with pytest.raises(pkg.capi.UvrtError, match="no batch of the same shape"):
            pkg.capi.reduce_batch_group([a, b])
        a.trace_batch(lamps[:2], 1.0, 0, n)
        with pytest.raises(pkg.capi.UvrtError, match="no communicator"):
            a.reduce_batch()
        with pytest.raises(pkg.capi.UvrtError, match="no batch of the same shape"):
            pkg.capi.reduce_batch_group([a, b])          # b holds none
        b.trace_batch(lamps, 1.0, 0, n)
        with pytest.raises(pkg.capi.UvrtError, match="no batch of the same shape"):
            pkg.capi.reduce_batch_group([a, b])          # 2 launches against 3
        # a single-context group with a communicator: the grouped RCCL branch (ncclGroupStart / AllReduce / GroupEnd)
        pkg.capi.comm_init_all([a])
        with pytest.raises(pkg.capi.UvrtError, match="communicator present"):
            pkg.capi.comm_init_all([a])
        with pytest.raises(pkg.capi.UvrtError, match="already has a communicator"):
            a.comm_init_rank(pkg.capi.comm_unique_id(), 0, 1)
        want = [a.read_batch_counts(k) for k in range(2)]
        pkg.capi.reduce_batch_group([a])
        a.reduce_batch()
        for k in range(2):
            assert np.array_equal(a.read_batch_counts(k), want[k])
        a.comm_destroy()
    finally:
        a.close()
        b.close()


def test_cu_reservation_for_the_collective_keeps_the_bits(pkg, orc, oscene, oroute):
    """While a communicator is set the launch lanes' streams carry a CU mask (one CU per XCD left to the context's stream
    for fold / all-reduce / replay, uvrt_capi_comm.hip set_lane_cu_mask) and the persistent grid shrinks to the CUs that are
    left: same counts and dose as without; the lanes get their plain streams back when the communicator goes."""
    lamps = [lamp_pos(orc, oscene, oroute, k) for k in (3, 4)]
    n = 400000
    a = pkg.capi.Ctx(0)
    b = pkg.capi.Ctx(0)
    try:
        for c in (a, b):
            c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
            c.reset(True)
        ops = make_ops(pkg, [60.0, 20.0], {1: 0}, n)
        a.trace_batch(lamps, 1.0, 0, n)
        want = [a.read_batch_counts(k) for k in range(2)]
        a.replay_batch(ops)
        b.comm_init_rank(pkg.capi.comm_unique_id(), 0, 1)             # lanes now masked
        for rnd in range(2):                                          # twice: both buffer sets, both lanes
            b.reset(True)
            b.seed = 0
            b.trace_batch(lamps, 1.0, 0, n)
            b.reduce_batch()
            for k in range(2):
                assert np.array_equal(b.read_batch_counts(k), want[k]), (rnd, k)
            b.replay_batch(ops)
            assert np.array_equal(bits(a.read_dosage()), bits(b.read_dosage()))
        # the per-launch path on the masked lanes
        rays, _ = orc.generate(0, n, lamps[0], 1.0, 0)
        temp = np.zeros(oscene.T, dtype=np.int32)
        orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
        for _ in range(3):                                            # lanes 0, 1, 0 ...
            b.resize_rays(n)
            b.reset(False)
            b.seed = 0
            b.generate(lamps[0], 1.0, 0, n)
            b.extend(n)
            assert np.array_equal(b.read_counts(), temp)
        b.comm_destroy()                                              # plain streams again
        b.reset(True)
        b.seed = 0
        b.trace_batch(lamps, 1.0, 0, n)
        b.replay_batch(ops)
        assert np.array_equal(bits(a.read_dosage()), bits(b.read_dosage())) and a.read_dosage().any()
    finally:
        a.close()
        b.close()
