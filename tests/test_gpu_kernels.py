"""GPU parity tests proper: every HIP kernel, called through the C ABI, against the CPU oracle
on the same seeded inputs.  Bar: bit-exact (rays, hit distances, triangle ids, integer photon
counts, f64 maps, f32 dose and colours)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg, oscene):
    c = pkg.capi.Ctx(0)
    c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx_dev(pkg, oscene):
    """A context of the developer build (libuvrt_hip_dev.so): the kernel knobs the product library does not hold."""
    c = pkg.capi.Ctx(0, dev=True)
    c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
    yield c
    c.close()


def lamp_pos(orc, oscene, oroute, k):
    comp = orc.Computation(oscene, oroute["lamps"], 1 << 16, oroute["lightHeight"],
                           oroute["lightLength"], oroute["lightIntensity"])
    return tuple(float(x) for x in comp.lamp_world_pos(oroute["lamps"][k]))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("n,first", [(1, 0), (63, 0), (4096, 0), (100000, 0), (5000, 777), (1, 12345)])
def test_generate_bit_exact(ctx, orc, oscene, oroute, n, first):
    lp = lamp_pos(orc, oscene, oroute, 0)
    ctx.resize_rays(max(n, 1))
    for seed0 in (0, 0x79044923):
        ctx.seed = seed0
        ctx.generate(lp, oroute["lightLength"], first, n)
        got = ctx.read_rays(0, n)
        ref, seed1 = orc.generate(first, n, lp, oroute["lightLength"], seed0)
        assert ctx.seed == seed1
        for f in ("dirx", "diry", "dirz", "origx", "origy", "origz", "dist"):
            assert np.array_equal(bits(got[f]), bits(ref[f])), f
        assert np.array_equal(got["triID"], ref["triID"])


def test_seed_chain_matches_survey(pkg):
    # SURVEY.md 8c "Pinned SEED semantics"
    s1 = pkg.capi.seed_next((-0.25500134, -0.99548361, -3.3149862), 1.0, 0)
    assert s1 == 0x79044923
    s2 = pkg.capi.seed_next((-1.1050029, -0.99548361, -3.2301083), 1.0, s1)
    assert s2 == 0xce0db3eb


@pytest.mark.parametrize("sort_bits", [0, 6, 12, -1])
def test_extend_hits_and_counts_bit_exact(ctx, orc, oscene, oroute, sort_bits):
    n = 200000
    lp = lamp_pos(orc, oscene, oroute, 3)
    ctx.set_sort_bits(sort_bits)
    ctx.set_record_hits(True)
    ctx.resize_rays(n)
    ctx.reset(False)
    ctx.seed = 0
    ctx.generate(lp, oroute["lightLength"], 0, n)
    ctx.extend(n)
    ctx.sync()
    got = ctx.read_rays(0, n)
    counts = ctx.read_counts()
    rays, _ = orc.generate(0, n, lp, oroute["lightLength"], 0)
    temp = np.zeros(oscene.T, dtype=np.int32)
    st = orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
    assert st["hits"] > 0.5 * n
    assert np.array_equal(bits(got["dist"]), bits(rays["dist"]))
    assert np.array_equal(got["triID"], rays["triID"])
    assert np.array_equal(counts, temp)
    ctx.set_sort_bits(-1)
    ctx.set_record_hits(False)


@pytest.mark.parametrize("merge", ["0", "1"])
@pytest.mark.parametrize("flavour", [0, 2])
def test_drain_merge_changes_nothing(pkg, orc, oscene, oroute, monkeypatch, merge, flavour):
    """k_extend6's workgroups pool the last rays of their four waves in one wave (merge6; by default only when launches can
    overlap).  Forced on and off (UVRT_DRAIN_MERGE, read at uvrt_create): per-ray (dist, triID) and counts are the oracle's either
    way -- launches of less than one batch per wave (waves without rays meet the barriers too), ragged sizes, several batches
    per wave, with and without hit records."""
    monkeypatch.setenv("UVRT_DRAIN_MERGE", merge)
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        c.set_flavour(flavour)
        lp = lamp_pos(orc, oscene, oroute, 5)
        orc.set_flavour(flavour)
        for n, rec in ((1, True), (200000, True), (200000, False), (458753, True), (1500001, False)):
            c.set_record_hits(rec)
            c.resize_rays(n)
            c.reset(False)
            c.seed = 0
            c.generate(lp, oroute["lightLength"], 0, n)
            c.extend(n)
            c.sync()
            rays, _ = orc.generate(0, n, lp, oroute["lightLength"], 0)
            temp = np.zeros(oscene.T, dtype=np.int32)
            orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
            assert np.array_equal(c.read_counts(), temp), (n, rec)
            if rec:
                got = c.read_rays(0, n)
                assert np.array_equal(bits(got["dist"]), bits(rays["dist"])), n
                assert np.array_equal(got["triID"], rays["triID"]), n
    finally:
        orc.set_flavour(0)
        c.close()


@pytest.mark.parametrize("variant", [0, 400, 401, 402, 403, 404, 405, 406, 407, 411, 421, 431, 441, 500, 501, 505, 601, 702, 801])
def test_every_extend_variant_is_bit_exact(pkg, ctx, ctx_dev, orc, oscene, oroute, variant):
    """All kernel knob settings (leaf period, LDS top cache on / off, 2-16 workgroups per CU, refill
    thresholds, IEEE divisions everywhere) give the
    same bits: layout and scheduling never change results.
    Knobs that select another instantiation than the product's run on the developer build of the library."""
    _variant_case(ctx_dev if pkg.capi.needs_dev(variant) else ctx, orc, oscene, oroute, variant, 0)


@pytest.mark.parametrize("variant", [0, 400, 401, 403, 404, 405, 421, 500, 501, 505, 601, 801])
def test_every_v6_variant_is_bit_exact_in_the_ocl_flavour(pkg, ctx, ctx_dev, orc, oscene, oroute, variant):
    """flavour 1 (fused cross/dot of ROCm's OpenCL library, uvrt_set_flavour) on the default kernel and
    its variants (leaf period, LDS top cache, grid, refill threshold, IEEE divisions) against the
    oracle in the same flavour."""
    _variant_case(ctx_dev if pkg.capi.needs_dev(variant) else ctx, orc, oscene, oroute, variant, 1)


def test_product_library_holds_the_default_kernel_only(pkg, ctx, ctx_dev):
    """Kernel knobs that select another instantiation (leaf periods 1 / 3 / 4, no LDS cache) exist in the developer
    build only; grid and refill knobs of the default kernel are in both."""
    for v in (400, 402, 405, 500, 606):
        assert pkg.capi.needs_dev(v)
        with pytest.raises(pkg.capi.UvrtError, match="developer build"):
            ctx.set_variant(v)
        ctx_dev.set_variant(v)
    for v in (401, 451, 501, 601, 801, 901, 1251):
        assert not pkg.capi.needs_dev(v)
        ctx.set_variant(v)
        ctx_dev.set_variant(v)
    for v in (1, 399, 1300, 2208):
        for c in (ctx, ctx_dev):
            with pytest.raises(pkg.capi.UvrtError):
                c.set_variant(v)
    ctx.set_variant(0)
    ctx_dev.set_variant(0)


def _variant_case(ctx, orc, oscene, oroute, variant, flavour):
    n = 300000
    lp = lamp_pos(orc, oscene, oroute, 5)
    ctx.set_variant(variant)
    ctx.set_flavour(flavour)
    ctx.set_sort_bits(0)
    ctx.set_record_hits(True)
    orc.set_flavour(flavour)
    try:
        ctx.resize_rays(n)
        ctx.reset(False)
        ctx.seed = 7
        ctx.generate(lp, oroute["lightLength"], 0, n)
        ctx.extend(n)
        ctx.sync()
        got = ctx.read_rays(0, n)
        counts = ctx.read_counts()
        rays, _ = orc.generate(0, n, lp, oroute["lightLength"], 7)
        temp = np.zeros(oscene.T, dtype=np.int32)
        orc.extend(temp, oscene.tris, rays, oscene.nodes, oscene.triIdx)
        assert np.array_equal(bits(got["dist"]), bits(rays["dist"]))
        assert np.array_equal(got["triID"], rays["triID"])
        assert np.array_equal(counts, temp)
    finally:
        orc.set_flavour(0)
        ctx.set_variant(0)
        ctx.set_flavour(0)
        ctx.set_record_hits(False)


def test_full_iteration_two_lamps_matches_survey_golden(ctx, orc, oscene, oroute):
    """generate -> extend -> accumulate per lamp, then computeDosage: SURVEY.md 8c golden run
    (N = 65 536, lamps 0 and 1) and the oracle, bit for bit."""
    import json, os
    from conftest import GOLDEN
    gold = json.load(open(os.path.join(GOLDEN, "survey_8c.json")))["runs"][0]
    lamps = oroute["lamps"][:2]
    comp = orc.Computation(oscene, lamps, gold["photonCount"], oroute["lightHeight"],
                           oroute["lightLength"], oroute["lightIntensity"])
    comp.reset()
    comp.iteration()
    ref_dose = comp.dose()
    ppl = comp.photonsPerLight
    ctx.resize_rays(ppl)
    ctx.reset(True)
    ctx.seed = 0
    size = 0
    for lamp in lamps:
        lp = comp.lamp_world_pos(lamp)
        ctx.generate(lp, oroute["lightLength"], 0, ppl)
        ctx.extend(ppl)
        ctx.accumulate(lamp[2])
        size += ppl
    scaled = np.float32(np.float32(oroute["lightIntensity"]) * np.float32(0.1))
    ctx.compute_dosage(pkg_map_sum(), size // len(lamps), scaled)
    ctx.sync()
    dose = ctx.read_dosage()
    assert np.array_equal(ctx.read_photon_map(0), comp.photonMap)
    assert np.array_equal(ctx.read_photon_map(1), comp.maxPhotonMap)
    assert np.array_equal(bits(dose), bits(ref_dose))
    assert np.allclose(dose[:8], gold["dose_0_7"], rtol=1e-7)
    assert abs(float(dose.astype(np.float64).sum()) - gold["dose_sum"]) < 0.01
    assert int((dose != 0).sum()) == gold["dose_nonzero"]
    assert ctx.seed == 0xce0db3eb
    # max-power view + colours
    ctx.compute_dosage(1, ppl, np.float32(np.float32(oroute["lightIntensity"]) * np.float32(100)))
    ctx.dosage_to_color(oroute["minPower"], False)
    ctx.sync()
    mp = ctx.read_dosage()
    assert np.array_equal(bits(mp), bits(comp.max_power()))
    col = ctx.read_color()
    assert np.array_equal(bits(col), bits(orc.dosage_to_color(mp, oroute["minPower"], False)))
    ctx.dosage_to_color(oroute["minPower"], True)
    ctx.sync()
    assert np.array_equal(bits(ctx.read_color()), bits(orc.dosage_to_color(mp, oroute["minPower"], True)))
    # the fused Shade launch (uvrt_shade) = computeDosage followed by dosageToColor, same bits
    for which, n_ppl, power, min_value, thr, want in (
            (0, size // len(lamps), scaled, oroute["minDosage"], False, ref_dose),
            (1, ppl, np.float32(np.float32(oroute["lightIntensity"]) * np.float32(100)), oroute["minPower"], True, mp)):
        ctx.reset(True)            # clears the colour buffer; the maps are rebuilt below
        ctx.seed = 0
        for lamp in lamps:
            ctx.generate(comp.lamp_world_pos(lamp), oroute["lightLength"], 0, ppl)
            ctx.extend(ppl)
            ctx.accumulate(lamp[2])
        ctx.shade(which, n_ppl, power, min_value, thr)
        ctx.sync()
        got = ctx.read_dosage()
        assert np.array_equal(bits(got), bits(want))
        assert np.array_equal(bits(ctx.read_color()), bits(orc.dosage_to_color(want, min_value, thr)))


def pkg_map_sum():
    return 0


def test_reset_clears_maps(ctx):
    ctx.reset(True)
    ctx.sync()
    assert not ctx.read_photon_map(0).any()
    assert not ctx.read_photon_map(1).any()
    assert not ctx.read_counts().any()
    assert not ctx.read_color().any()


def test_root_leaf_scene(pkg, orc):
    """CalibratePower's scene: two triangles under a single leaf root (raytracer.cpp:158-187)."""
    tris = np.zeros((2, 16), dtype=np.float32)
    h, d, w = np.float32(-0.5), np.float32(1.0), np.float32(0.1)
    tris[0, 0:3] = (w, h + w, d); tris[0, 4:7] = (-w, h + w, d); tris[0, 8:11] = (w, h - w, d)
    tris[1, 0:3] = (-w, h - w, d); tris[1, 4:7] = (-w, h + w, d); tris[1, 8:11] = (w, h - w, d)
    nodes = np.zeros(1, dtype=orc.NODE_DT)
    nodes[0]["leftFirst"] = 0
    nodes[0]["triCount"] = 2
    idx = np.array([0, 1], dtype=np.uint32)
    c = pkg.capi.Ctx(0)
    try:
        c.set_scene(tris, nodes, idx)
        n = 300000
        c.resize_rays(n)
        c.set_record_hits(True)
        c.reset(False)
        lp = (0.0, -1.0, 0.0)
        c.generate(lp, 1.0, 0, n)
        c.extend(n)
        c.sync()
        got = c.read_rays(0, n)
        counts = c.read_counts()
        rays, _ = orc.generate(0, n, lp, 1.0, 0)
        temp = np.zeros(2, dtype=np.int32)
        orc.extend(temp, tris, rays, nodes, idx)
        assert temp.sum() > 0
        assert np.array_equal(counts, temp)
        assert np.array_equal(bits(got["dist"]), bits(rays["dist"]))
        assert np.array_equal(got["triID"], rays["triID"])
    finally:
        c.close()


def test_no_scene_is_an_error(pkg):
    c = pkg.capi.Ctx(0)
    try:
        with pytest.raises(pkg.capi.UvrtError):
            c.reset(True)
        c.resize_rays(16)
        c.generate((0, 0, 0), 1.0, 0, 16)
        with pytest.raises(pkg.capi.UvrtError):
            c.extend(16)
        with pytest.raises(pkg.capi.UvrtError):
            c.generate((0, 0, 0), 1.0, 0, 17)   # beyond capacity
    finally:
        c.close()


def test_ray_range_sharding_equals_whole_launch(pkg, orc, oscene, oroute):
    """BASELINE configs[3] ("pixel tile"): two contexts trace disjoint gid ranges of ONE launch
    (global id in the seed, same SEED pair); their int counts add up to the unsharded launch."""
    n = 500000
    lp = lamp_pos(orc, oscene, oroute, 2)
    whole = pkg.capi.Ctx(0)
    whole.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
    whole.resize_rays(n)
    whole.reset(False)
    whole.seed = 99
    whole.generate(lp, oroute["lightLength"], 0, n)
    whole.extend(n)
    whole.sync()
    ref = whole.read_counts()
    seed_after = whole.seed
    total = np.zeros_like(ref)
    cuts = [0, 123457, n]
    for a, b in zip(cuts[:-1], cuts[1:]):
        c = pkg.capi.Ctx(0)
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        c.resize_rays(b - a)
        c.reset(False)
        c.seed = 99
        c.generate(lp, oroute["lightLength"], a, b - a)
        c.extend(b - a)
        c.sync()
        assert c.seed == seed_after          # every shard advances SEED identically
        total += c.read_counts()
        c.close()
    whole.close()
    assert np.array_equal(total, ref) and ref.sum() > 0.5 * n


def test_run_to_run_determinism(ctx, orc, oscene, oroute):
    """Deposits are integer atomics and every other kernel is element-wise: two runs of the same
    computation give identical counts, maps and dose bits whatever the scheduling."""
    lp = lamp_pos(orc, oscene, oroute, 7)
    n = 400000
    out = []
    ctx.set_variant(0)
    ctx.set_sort_bits(0)
    ctx.resize_rays(n)
    for rep in range(3):
        ctx.reset(True)
        ctx.seed = 5
        for _ in range(2):
            ctx.generate(lp, oroute["lightLength"], 0, n)
            ctx.extend(n)
            ctx.accumulate(60.0)
        ctx.compute_dosage(0, n, 44.0)
        ctx.sync()
        out.append((ctx.read_photon_map(0), ctx.read_photon_map(1), bits(ctx.read_dosage()), ctx.seed))
    for o in out[1:]:
        assert np.array_equal(o[0], out[0][0]) and np.array_equal(o[1], out[0][1])
        assert np.array_equal(o[2], out[0][2]) and o[3] == out[0][3]


def test_abi_robustness_sequences(pkg, orc, oscene, oroute):
    """Call sequences a host can legally produce: empty launches, scene swaps to a different
    triangle count and back (CalibratePower), ray-buffer resizes, knobs toggled between launches."""
    c = pkg.capi.Ctx(0)
    try:
        assert c.device_cus() > 0
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        lp = lamp_pos(orc, oscene, oroute, 1)
        c.resize_rays(0)
        c.generate(lp, 1.0, 0, 0)          # empty launch: nothing traced, SEED still advances
        c.extend(0)
        c.accumulate(60.0)
        c.sync()
        _, s1 = orc.generate(0, 1, lp, 1.0, 0)
        assert c.seed == s1 and not c.read_photon_map(0).any()

        def run(n, seed):
            c.resize_rays(n)
            c.reset(True)
            c.seed = seed
            c.generate(lp, oroute["lightLength"], 0, n)
            c.extend(n)
            c.sync()
            return c.read_counts()

        def ref(n, seed, tris, nodes, idx):
            rays, _ = orc.generate(0, n, lp, oroute["lightLength"], seed)
            t = np.zeros(tris.shape[0], dtype=np.int32)
            orc.extend(t, tris, rays, nodes, idx)
            return t

        a = run(70000, 3)
        assert np.array_equal(a, ref(70000, 3, oscene.tris, oscene.nodes, oscene.triIdx))
        # swap to a tiny scene (different T: maps are reallocated), trace, swap back
        tris2 = np.zeros((2, 16), dtype=np.float32)
        tris2[0, 0:3] = (1, -2, 2); tris2[0, 4:7] = (-1, -2, 2); tris2[0, 8:11] = (1, 1, 2)
        tris2[1, 0:3] = (-1, 1, 2); tris2[1, 4:7] = (-1, -2, 2); tris2[1, 8:11] = (1, 1, 2)
        nodes2 = np.zeros(1, dtype=orc.NODE_DT); nodes2[0]["triCount"] = 2
        idx2 = np.array([0, 1], dtype=np.uint32)
        c.set_scene(tris2, nodes2, idx2)
        b = run(5000, 4)
        assert b.shape == (2,) and np.array_equal(b, ref(5000, 4, tris2, nodes2, idx2))
        c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
        # knobs between launches, larger buffer after a smaller one
        c.set_sort_bits(10)
        c.set_record_hits(True)
        d = run(130000, 5)
        assert np.array_equal(d, ref(130000, 5, oscene.tris, oscene.nodes, oscene.triIdx))
        got = c.read_rays(100, 50)
        rays, _ = orc.generate(0, 130000, lp, oroute["lightLength"], 5)
        assert np.array_equal(bits(got["dirx"]), bits(rays["dirx"][100:150]))
        c.set_sort_bits(0)
        c.set_record_hits(False)
        c.set_timing(True)
        run(64, 6)
        ms, k = c.extend_time_ms()
        assert k == 1 and ms > 0
        c.set_timing(False)
        with pytest.raises(pkg.capi.UvrtError):
            c.read_rays(0, 65)             # beyond the last launch
        with pytest.raises(pkg.capi.UvrtError):
            c.read_dosage(0, oscene.T + 1)
        with pytest.raises(pkg.capi.UvrtError):
            c.extend(63)                   # does not match the last generate
    finally:
        c.close()
