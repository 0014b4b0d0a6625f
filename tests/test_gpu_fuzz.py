"""GPU: randomised differential test -- small random scenes (the oracle's BVH builder: bvh.cpp's order), random lamp
positions, ray counts that are not multiples of 64, the three arithmetic flavours and both SEED semantics, through the three
ways the product traces (per-launch calls with and without launch pipelining, batched tracing with its fused launches).
Everything against the oracle, bit for bit: per-ray (dist, triID) where the path records them, counts, maps, dose.  The
generator is seeded: a failure names its case."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def random_scene(rng):
    """triangles of mixed sizes in a box around the origin; sometimes clustered (deep, unbalanced trees), sometimes with
    duplicated triangles (exactly tied hits: the traversal order decides, extend.cl:25)"""
    T = int(rng.choice([3, 17, 200, 1500, 6000]))
    extent = float(rng.choice([0.5, 2.0, 10.0]))
    size = float(rng.choice([0.01, 0.1, 0.6])) * extent
    ctr = rng.uniform(-extent, extent, (T, 1, 3))
    if rng.random() < 0.4:                       # clusters
        k = max(1, T // 50)
        ctr = ctr[rng.integers(0, k, T)] + rng.normal(scale=0.02 * extent, size=(T, 1, 3))
    tris = np.zeros((T, 16), dtype=np.float32)
    tris[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]] = (ctr + rng.normal(scale=size, size=(T, 3, 3))).reshape(T, 9).astype(np.float32)
    if T >= 17 and rng.random() < 0.3:           # duplicates
        d = rng.integers(0, T, T // 8)
        tris[rng.integers(0, T, T // 8)] = tris[d]
    return tris, extent


import os

# UVRT_FUZZ_CASES=N widens the sweep (a soak run: 600 cases take about a minute on the GPU box)
CASES = list(range(int(os.environ.get("UVRT_FUZZ_CASES", "64"))))


@pytest.mark.parametrize("case", CASES)
def test_random_scene_lamp_and_launch_shape(pkg, orc, case):
    rng = np.random.default_rng(1000 + case)
    tris, extent = random_scene(rng)
    T = tris.shape[0]
    nodes, idx = orc.build_bvh(tris)
    flavour = int(rng.integers(0, 2))
    if case % 4 == 3:
        # the "shipped flags" arithmetic (uvrt_set_flavour 2): the oracle follows through its model of v_rcp_f32, whose table
        # is read from this GPU (oracle/rcp_model.h; checked exhaustively in test_gpu_shipped_flags.py)
        flavour = 2
        if orc.refgpu() is None:
            flavour = 1
    seed_mode = int(rng.integers(0, 2))
    n = int(rng.choice([1, 63, 64, 65, 1000, 20001, 70000]))
    launches = int(rng.integers(1, 5))
    lamps = [tuple(float(np.float32(v)) for v in rng.uniform(-0.6 * extent, 0.6 * extent, 3)) for _ in range(launches)]
    if rng.random() < 0.3:
        lamps = [lamps[0]] * launches                      # one lamp column: fused launches in batched mode
    length = float(np.float32(rng.choice([0.0, 0.5, 2.0]) * extent))
    durations = [float(np.float32(rng.uniform(0.5, 90.0))) for _ in range(launches)]
    seed0 = int(rng.integers(0, 2**32))

    # oracle
    orc.set_flavour(flavour)
    try:
        pm, mm = np.zeros(T), np.zeros(T)
        temp = np.zeros(T, dtype=np.int32)
        seed = seed0
        o_rays, o_counts = [], []
        for k in range(launches):
            if seed_mode == 0:
                rays, seed = orc.generate(0, n, lamps[k], length, seed)
            else:
                rays, seed = orc.generate_fixed_seed(0, n, lamps[k], length, seed, saturate=True)
            orc.extend(temp, tris, rays, nodes, idx)
            o_rays.append(rays)
            o_counts.append(temp.copy())
            orc.accumulate(pm, mm, temp, durations[k])
        dose = orc.compute_dosage(pm, tris, launches * n, np.float32(3.5))
    finally:
        orc.set_flavour(0)

    for how in ("calls", "calls_one_stream_recorded", "batched"):
        c = pkg.capi.Ctx(0)
        try:
            c.set_scene(tris, nodes, idx)
            c.set_flavour(flavour)
            c.set_seed_mode(seed_mode)
            c.resize_rays(n)
            c.reset(True)
            c.seed = seed0
            if how == "batched":
                c.trace_batch(lamps, length, 0, n)
                for k in range(launches):
                    assert np.array_equal(c.read_batch_counts(k), o_counts[k]), (case, how, k)
                ops = np.zeros(launches, dtype=pkg.capi.REPLAY_OP_DT)
                for k in range(launches):
                    ops[k] = (durations[k], 1 if k == launches - 1 else 0, 0, launches * n, 3.5, 1.0, 0)
                c.replay_batch(ops)
            else:
                if how == "calls_one_stream_recorded":
                    c.set_pipeline(False)
                    c.set_record_hits(True)
                for k in range(launches):
                    c.generate(lamps[k], length, 0, n)
                    c.extend(n)
                    if how == "calls_one_stream_recorded":
                        got = c.read_rays(0, n)
                        for f in ("dirx", "diry", "dirz", "origx", "origy", "origz", "dist"):
                            assert np.array_equal(bits(got[f]), bits(o_rays[k][f])), (case, how, k, f)
                        assert np.array_equal(got["triID"], o_rays[k]["triID"]), (case, how, k)
                        assert np.array_equal(c.read_counts(), o_counts[k]), (case, how, k)
                    c.accumulate(durations[k])
                c.compute_dosage(0, launches * n, 3.5)
            c.sync()
            assert c.seed == seed, (case, how)
            assert np.array_equal(c.read_photon_map(0), pm) and np.array_equal(c.read_photon_map(1), mm), (case, how)
            assert np.array_equal(bits(c.read_dosage()), bits(dose)), (case, how)
        finally:
            c.close()
