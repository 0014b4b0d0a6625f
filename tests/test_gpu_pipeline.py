"""GPU: launch pipelining (include/uvrt.h uvrt_set_pipeline).  Consecutive launches alternate between
two streams and two sets of ray / count buffers; whatever the interleaving of ABI calls, every
result must equal the one-stream behaviour and the oracle, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def ctx(pkg, oscene):
    c = pkg.capi.Ctx(0)
    c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
    yield c
    c.close()


def positions(orc, oscene, oroute):
    comp = orc.Computation(oscene, oroute["lamps"], 1 << 16, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    return [tuple(float(x) for x in comp.lamp_world_pos(l)) for l in oroute["lamps"]]


def run_sequence(ctx, oroute, lps, n, launches, pipeline, peek=None, toggle_at=None):
    """`launches` = list of (lamp index, duration).  peek(k) is called between extend and accumulate."""
    ctx.set_pipeline(pipeline)
    ctx.resize_rays(n)
    ctx.reset(True)
    ctx.seed = 0
    for k, (li, dur) in enumerate(launches):
        if toggle_at is not None and k == toggle_at:
            ctx.set_pipeline(not pipeline)
        ctx.generate(lps[li], oroute["lightLength"], 0, n)
        ctx.extend(n)
        if peek is not None:
            peek(k)
        ctx.accumulate(dur)
        ctx.shade(0, n * (k + 1) // 1, np.float32(44.0), oroute["minDosage"], False)
    ctx.sync()
    out = dict(sum=ctx.read_photon_map(0), max=ctx.read_photon_map(1), dose=ctx.read_dosage(),
               color=ctx.read_color(), seed=ctx.seed)
    ctx.set_pipeline(True)
    return out


def same(a, b):
    return (np.array_equal(a["sum"], b["sum"]) and np.array_equal(a["max"], b["max"]) and
            np.array_equal(bits(a["dose"]), bits(b["dose"])) and np.array_equal(bits(a["color"]), bits(b["color"]))
            and a["seed"] == b["seed"])


def test_pipelined_equals_one_stream_equals_oracle(ctx, orc, oscene, oroute):
    lps = positions(orc, oscene, oroute)
    n = 150000
    launches = [(0, 60.0), (5, 10.0), (11, 35.0), (0, 60.0), (7, 1.0), (3, 20.0), (3, 20.0)]
    a = run_sequence(ctx, oroute, lps, n, launches, True)
    b = run_sequence(ctx, oroute, lps, n, launches, False)
    c = run_sequence(ctx, oroute, lps, n, launches, True, toggle_at=3)
    assert same(a, b) and same(a, c)
    # the oracle, launch by launch
    pm = np.zeros(oscene.T, dtype=np.float64)
    mm = np.zeros(oscene.T, dtype=np.float64)
    seed = 0
    for li, dur in launches:
        rays, seed = orc.generate(0, n, lps[li], oroute["lightLength"], seed)
        temp = np.zeros(oscene.T, dtype=np.int32)
        orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
        orc.accumulate(pm, mm, temp, dur)
    assert np.array_equal(a["sum"], pm) and np.array_equal(a["max"], mm) and a["seed"] == seed


def test_reads_between_the_calls_of_a_launch(ctx, orc, oscene, oroute):
    """read_counts between extend and accumulate (on alternating lanes) returns that launch's
    counts and does not disturb the pipeline."""
    lps = positions(orc, oscene, oroute)
    n = 60000
    launches = [(2, 5.0), (9, 5.0), (4, 7.0), (6, 9.0)]
    seen = []
    a = run_sequence(ctx, oroute, lps, n, launches, True, peek=lambda k: seen.append(ctx.read_counts()))
    b = run_sequence(ctx, oroute, lps, n, launches, False)
    assert same(a, b)
    seed = 0
    for k, (li, _) in enumerate(launches):
        rays, seed = orc.generate(0, n, lps[li], oroute["lightLength"], seed)
        temp = np.zeros(oscene.T, dtype=np.int32)
        orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
        assert np.array_equal(seen[k], temp), k


def test_reset_and_rescene_in_the_middle(ctx, pkg, orc, oscene, oroute):
    """reset, a scene swap to the two-triangle calibration scene and back (raytracer.cpp:166-224) and
    a resize between pipelined launches."""
    lps = positions(orc, oscene, oroute)
    n = 40000
    ref = run_sequence(ctx, oroute, lps, n, [(1, 3.0), (8, 4.0)], False)
    ctx.set_pipeline(True)
    ctx.resize_rays(n)
    ctx.reset(True)
    ctx.seed = 99
    for li in (4, 6, 10):                           # work that must leave no trace
        ctx.generate(lps[li], oroute["lightLength"], 0, n)
        ctx.extend(n)
        ctx.accumulate(1.0)
    # calibration scene: one square of two triangles, root-leaf BVH
    tris = np.zeros((2, 16), dtype=np.float32)
    tris[0, [0, 1, 2]] = (-1, 0, -1); tris[0, [4, 5, 6]] = (1, 0, -1); tris[0, [8, 9, 10]] = (1, 0, 1)
    tris[1, [0, 1, 2]] = (-1, 0, -1); tris[1, [4, 5, 6]] = (1, 0, 1); tris[1, [8, 9, 10]] = (-1, 0, 1)
    nodes = np.zeros(1, dtype=orc.NODE_DT)
    nodes[0] = (-1, 0, -1, 0, 1, 0, 1, 2)
    ctx.set_scene(tris, nodes, np.array([0, 1], dtype=np.uint32))
    ctx.resize_rays(1000)
    ctx.generate((0.0, 0.5, 0.0), 1.0, 0, 1000)
    ctx.extend(1000)
    ctx.accumulate(1.0)
    ctx.sync()
    assert ctx.read_photon_map(0).sum() > 0
    ctx.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
    got = run_sequence(ctx, oroute, lps, n, [(1, 3.0), (8, 4.0)], True)
    assert same(got, ref)


def test_record_renumbering_does_not_change_results(ctx, orc, oscene, oroute):
    """uvrt_set_record_perm moves other node-pair records into the LDS-cached prefix of the default
    kernel; any permutation must give the same counts (here: a random one and the reversal)."""
    lps = positions(orc, oscene, oroute)
    n = 80000
    nodes = oscene.nodes                                     # inner nodes reachable from the root
    q = [0] if nodes[0]["triCount"] == 0 else []
    i = 0
    while i < len(q):
        l = int(nodes[q[i]]["leftFirst"]); i += 1
        q += [l + k for k in (0, 1) if nodes[l + k]["triCount"] == 0]
    npairs = len(q)
    ctx.set_pipeline(True)
    ctx.resize_rays(n)
    rays, _ = orc.generate(0, n, lps[0], oroute["lightLength"], 5)
    ref = np.zeros(oscene.T, dtype=np.int32)
    orc.extend(ref, oscene.tris, rays, oscene.nodes, oscene.triIdx)
    rng = np.random.default_rng(0)
    try:
        for perm in (rng.permutation(npairs), np.arange(npairs)[::-1], None):
            ctx.set_record_perm(perm)
            ctx.reset(False)
            ctx.seed = 5
            ctx.generate(lps[0], oroute["lightLength"], 0, n)
            ctx.extend(n)
            assert np.array_equal(ctx.read_counts(), ref)
            ctx.accumulate(1.0)
        with pytest.raises(RuntimeError):
            ctx.set_record_perm(np.zeros(npairs, dtype=np.uint32))      # not a permutation
    finally:
        ctx.set_record_perm(None)


def test_back_to_back_computations_without_host_sync(ctx, orc, oscene, oroute):
    """Several computations (reset + launches) enqueued back to back: the next one's first launches
    overlap the previous one's drain (map fence), yet each reset must see all earlier map updates
    finished and each accumulate the reset before it.  One computation ends with an extend that is
    never accumulated (dirty count buffers: the reset then fences everything)."""
    lps = positions(orc, oscene, oroute)
    n = 50000

    def computation(k, leave_dirty):
        ctx.reset(True)
        ctx.seed = k
        for j, li in enumerate((k % 12, (k + 5) % 12, (k + 7) % 12)):
            ctx.generate(lps[li], oroute["lightLength"], 0, n)
            ctx.extend(n)
            if leave_dirty and j == 2:
                break
            ctx.accumulate(10.0 + k)
            ctx.shade(0, n, np.float32(44.0), oroute["minDosage"], False)

    results = {}
    for pipeline in (True, False):
        ctx.set_pipeline(pipeline)
        ctx.resize_rays(n)
        for k in range(5):
            computation(k, leave_dirty=(k == 2))
        ctx.sync()
        results[pipeline] = dict(sum=ctx.read_photon_map(0), max=ctx.read_photon_map(1), dose=ctx.read_dosage(),
                                 color=ctx.read_color(), seed=ctx.seed)
    ctx.set_pipeline(True)
    assert same(results[True], results[False])
    # the oracle for the last computation (k = 4)
    pm = np.zeros(oscene.T, dtype=np.float64)
    mm = np.zeros(oscene.T, dtype=np.float64)
    seed = 4
    for li in (4, 9, 11):
        rays, seed = orc.generate(0, n, lps[li], oroute["lightLength"], seed)
        temp = np.zeros(oscene.T, dtype=np.int32)
        orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
        orc.accumulate(pm, mm, temp, 14.0)
    assert np.array_equal(results[True]["sum"], pm) and np.array_equal(results[True]["max"], mm)


def test_external_work_on_the_maps_is_ordered_after_all_lanes(pkg, orc, oscene, oroute):
    """What sharding.MapReducer relies on: after uvrt_device_ptr, work enqueued on the context's
    stream (here: a torch copy of photonMap on the torch stream the context runs on) sees every
    launch, including those still running on the library's second stream; and an accumulate that
    follows waits for that external work (the copy is overwritten with a marker first)."""
    import torch
    from uvrt_amd import sharding
    lps = positions(orc, oscene, oroute)
    n = 400000
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        c.set_stream(stream.cuda_stream)
        c.resize_rays(n)
        with torch.cuda.stream(stream):
            reducer = sharding.MapReducer(c, dev)
            c.reset(False)
            c.seed = 0
            for li in (0, 3, 6):                      # three launches: the last sits on a side lane
                c.generate(lps[li], oroute["lightLength"], 0, n)
                c.extend(n)
                c.accumulate(2.0)
            reducer()                                  # world size 1: no collective, but the ordering call
            snap = reducer.sum_t.clone()               # external work on the context's stream
            reducer.sum_t.mul_(2.0)                    # ... that also MODIFIES the map
            c.generate(lps[9], oroute["lightLength"], 0, n)
            c.extend(n)
            c.accumulate(2.0)                          # must come after the doubling
            c.sync()
            final = c.read_photon_map(0)
        torch.cuda.synchronize()
        pm = np.zeros(oscene.T, dtype=np.float64)
        mm = np.zeros(oscene.T, dtype=np.float64)
        seed = 0
        for li in (0, 3, 6):
            rays, seed = orc.generate(0, n, lps[li], oroute["lightLength"], seed)
            temp = np.zeros(oscene.T, dtype=np.int32)
            orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
            orc.accumulate(pm, mm, temp, 2.0)
        assert np.array_equal(snap.cpu().numpy(), pm)
        rays, seed = orc.generate(0, n, lps[9], oroute["lightLength"], seed)
        temp = np.zeros(oscene.T, dtype=np.int32)
        orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
        want = pm * 2.0
        orc.accumulate(want, mm, temp, 2.0)
        assert np.array_equal(final, want)
    finally:
        c.close()


def test_the_deferred_accumulate_that_a_shade_takes_along_changes_nothing(pkg, orc, oscene, oroute):
    """A full-range uvrt_accumulate is enqueued with the NEXT call: a uvrt_shade right behind it (the host loop's order,
    myapp.cpp:159-160) runs accumulate + computeDosage + dosageToColor in one kernel, anything else launches the accumulate first.
    Maps, dose, colours and SEED equal the oracle's and the one-stream sequence's whatever comes between the two calls."""
    lps = positions(orc, oscene, oroute)
    n = 70001
    length = oroute["lightLength"]
    plan = [(0, 0), (1, 1), (1, 0), (2, 1), (0, 0), (0, 1)]           # (lamp, which map the Shade reads)

    def run(between, pipeline=True):
        c = pkg.capi.Ctx(0)
        try:
            c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
            c.set_pipeline(pipeline)
            c.resize_rays(n)
            c.reset(True)
            c.seed = 99
            out = []
            for k, (li, which) in enumerate(plan):
                c.generate(lps[li], length, 0, n)
                c.extend(n)
                counts = c.read_counts() if k == 2 else None                # a test hook between extend and accumulate
                c.accumulate(10.0 + k)
                if between == "sync":
                    c.sync()
                elif between == "read":
                    c.read_photon_map(k & 1)
                elif between == "dose only":
                    c.compute_dosage(which, (k + 1) * n, 44.0)
                elif between == "twice" and k == 3:
                    c.shade(1 - which, 5, 1.0, 1.0, 0)                      # an extra Shade: the second one finds nothing pending
                c.shade(which, (k + 1) * n, 44.0, 100.0, k & 1)
                if k & 1:
                    c.sync()
                out.append((c.read_dosage(), c.read_color(), c.seed, counts))
            out.append((c.read_photon_map(0), c.read_photon_map(1)))
            return out
        finally:
            c.close()

    ref = run(None, pipeline=False)
    for between in (None, "sync", "read", "dose only", "twice"):
        got = run(between)
        for a, b in zip(ref[:-1], got[:-1]):
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), between
            assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), between
            assert a[2] == b[2], between
            assert (a[3] is None) or np.array_equal(a[3], b[3]), between
        assert np.array_equal(ref[-1][0], got[-1][0]) and np.array_equal(ref[-1][1], got[-1][1]), between
    # and against the oracle
    pm, mm, temp = np.zeros(oscene.T), np.zeros(oscene.T), np.zeros(oscene.T, dtype=np.int32)
    seed = 99
    for k, (li, _) in enumerate(plan):
        rays, seed = orc.generate(0, n, lps[li], length, seed)
        orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
        orc.accumulate(pm, mm, temp, 10.0 + k)
    assert np.array_equal(ref[-1][0], pm) and np.array_equal(ref[-1][1], mm) and ref[-2][2] == seed
    dose = orc.compute_dosage(mm, oscene.tris, 6 * n, np.float32(44.0))
    assert np.array_equal(ref[-2][0].view(np.uint32), dose.view(np.uint32))
    col = orc.dosage_to_color(dose, 100.0, True)
    assert np.array_equal(ref[-2][1].view(np.uint32), col.view(np.uint32))
