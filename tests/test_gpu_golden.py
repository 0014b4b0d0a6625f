"""GPU: the HIP path against the COMMITTED golden fixtures (tests/golden/oracle_small.npz, made
by tests/golden/make_golden.py from the pinned oracle) -- no CPU oracle run involved -- and
size-independent properties at the bench's full size."""
import json
import os

import numpy as np
import pytest

from conftest import GLB, GOLDEN, ROUTE

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def host(pkg):
    from uvrt_amd import host
    return host


def test_hip_path_reproduces_committed_fixture(host):
    z = np.load(os.path.join(GOLDEN, "oracle_small.npz"))
    rt = host.RayTracer(GLB, ROUTE, device=0)          # product loader + native BVH
    rt.set_lamps(rt.lamps()[:1])
    rt.photonCount = 65536
    rt.ctx.set_record_hits(True)
    rt.ResetDosageMap()
    lamp = rt.lamps()[0]
    c = rt.ctx
    lp = tuple(z["light_pos"])
    c.generate(lp, 1.0, 0, 65536)
    c.extend(65536)
    c.sync()
    assert c.seed == int(z["seed1"])
    rays = c.read_rays(0, 4096)
    g = z["rays256"]
    for f in ("dirx", "diry", "dirz", "origx", "origy", "origz"):
        assert np.array_equal(bits(rays[f][:256]), bits(g[f])), f
    assert np.array_equal(bits(rays["dist"]), bits(z["dist4096"]))
    assert np.array_equal(rays["triID"], z["tri4096"])
    assert np.array_equal(c.read_counts(), z["counts"])
    c.accumulate(lamp[2])
    c.compute_dosage(0, 65536, np.float32(np.float32(rt.lightIntensity) * np.float32(0.1)))
    c.sync()
    dose = c.read_dosage()
    assert np.array_equal(bits(dose[:2048]), bits(z["dose2048"]))
    assert float(dose.astype(np.float64).sum()) == float(z["dose_sum"])
    rt.close()


def test_full_size_properties(host):
    """BASELINE size (2 073 600 photons per launch): conservation, determinism across ray
    orderings, linearity of the sum map, SEED chain, and the census-derived hit count."""
    census = json.load(open(os.path.join(GOLDEN, "census_lamp0_2073600.json")))
    n = 2073600
    rt = host.RayTracer(GLB, ROUTE, device=0)
    rt.set_lamps(rt.lamps()[:1])
    rt.photonCount = n
    c = rt.ctx
    results = []
    for sort_bits in (0, -1):
        c.set_sort_bits(sort_bits)
        rt.ResetDosageMap()
        c.seed = 0
        rt.ComputeDosageMap()                    # launch 0
        c.sync()
        pm1 = c.read_photon_map(0)
        hits0 = int(round(pm1.sum() / 60.0))
        assert hits0 == census["per_launch"][0]["hits"]          # every hit deposited exactly once
        rt.ComputeDosageMap()                    # launch 1
        c.sync()
        pm2, mx2 = c.read_photon_map(0), c.read_photon_map(1)
        assert int(round(pm2.sum() / 60.0)) == hits0 + census["per_launch"][1]["hits"]
        assert (mx2 * 60.0 <= pm2).all() and (mx2 >= (pm2 - pm1) / 60.0).all()
        rt.Shade()
        c.sync()
        results.append((pm2, mx2, rt.read_dosage()))
    assert np.array_equal(results[0][0], results[1][0])           # ordering never changes results
    assert np.array_equal(results[0][1], results[1][1])
    assert np.array_equal(bits(results[0][2]), bits(results[1][2]))
    c.set_sort_bits(0)
    rt.close()
