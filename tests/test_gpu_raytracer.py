"""GPU: the RayTracer class (C++ host mirror on the HIP C ABI) against the oracle running the
reference's launch sequence, the calibration path (3.4) and launch sharding."""
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


def test_tick_loop_three_lamps_two_iterations(host, orc, oscene, oroute):
    rt = host.RayTracer(GLB, ROUTE, device=0)
    rt.set_lamps(rt.lamps()[:3])
    rt.photonCount = 300000
    rt.maxIterations = 2
    rt.ResetDosageMap()
    comp = orc.Computation(oscene, oroute["lamps"][:3], 300000, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    comp.reset()
    assert rt.photonsPerLight == comp.photonsPerLight == 100000
    for it in range(2):                     # myapp.cpp:156-175
        rt.ComputeDosageMap()
        rt.Shade()
        rt.currIterations = rt.currIterations + 1
        rt.Sync()
        comp.iteration()
        assert rt.photonMapSize == comp.photonMapSize
        assert np.array_equal(bits(rt.read_dosage()), bits(comp.dose()))
    assert np.array_equal(rt.ctx.read_photon_map(0), comp.photonMap)
    assert np.array_equal(rt.ctx.read_photon_map(1), comp.maxPhotonMap)
    rt.viewMode = host.VIEW_MAXPOWER
    rt.Shade()
    rt.Sync()
    mp = comp.max_power()
    assert np.array_equal(bits(rt.read_dosage()), bits(mp))
    assert np.array_equal(bits(rt.ctx.read_color()), bits(orc.dosage_to_color(mp, oroute["minPower"], False)))
    rt.close()


def oracle_calibrate(orc, floorHeight, lightHeight, lightLength, photonCount, maxIterations, measurePower,
                     measureHeight, measureDist, SEED):
    """raytracer.cpp:151-227 on the oracle kernels."""
    f = np.float32
    h = f(f(measureHeight) + f(floorHeight))
    w, d = f(0.1), f(f(0.0) + f(measureDist))
    tris = np.zeros((2, 16), dtype=np.float32)
    tris[0, 0:3] = (f(0) + w, h + w, d); tris[0, 4:7] = (f(0) - w, h + w, d); tris[0, 8:11] = (f(0) + w, h - w, d)
    tris[1, 0:3] = (f(0) - w, h - w, d); tris[1, 4:7] = (f(0) - w, h + w, d); tris[1, 8:11] = (f(0) + w, h - w, d)
    nodes = np.zeros(1, dtype=orc.NODE_DT)
    nodes[0]["triCount"] = 2
    idx = np.array([0, 1], dtype=np.uint32)
    pm, mm, temp = np.zeros(2), np.zeros(2), np.zeros(2, dtype=np.int32)
    lp = (f(0), f(f(floorHeight) + f(lightHeight)), f(0))
    for _ in range(maxIterations):
        rays, SEED = orc.generate(0, photonCount, lp, lightLength, SEED)
        orc.extend(temp, tris, rays, nodes, idx)
        orc.accumulate(pm, mm, temp, 0.0)
    dose = orc.compute_dosage(mm, tris, photonCount, 1.0)
    avg = f(f(dose[0] + dose[1]) / f(2.0))
    return f(f(0.01) * f(f(measurePower) / avg)), SEED, dose


def test_calibrate_power_then_compute(host, orc, oscene, oroute):
    rt = host.RayTracer(GLB, ROUTE, device=0)
    rt.set_lamps(rt.lamps()[:1])
    rt.photonCount = 400000
    rt.maxIterations = 3
    # colours on screen before the calibration (the reference keeps them: ClearBuffers(false), raytracer.cpp:187,224)
    rt.ResetDosageMap()
    rt.ComputeDosageMap()
    rt.Shade()
    colours = rt.ctx.read_color()
    assert colours.any()
    rt.ctx.seed = 0
    rt.CalibratePower(2909.0, 0.8, 1.0)      # userinterface.cpp:107-109 defaults
    assert np.array_equal(bits(rt.ctx.read_color()), bits(colours))
    power, seed, dose = oracle_calibrate(orc, oscene.floorHeight, oroute["lightHeight"], oroute["lightLength"],
                                         400000, 3, 2909.0, 0.8, 1.0, 0)
    assert dose.min() > 0
    assert np.float32(rt.lightIntensity) == power
    assert np.float32(rt.calibratedPower) == power
    assert rt.ctx.seed == seed
    # the room is back: a full computation afterwards still matches the oracle
    rt.ResetDosageMap()
    rt.ComputeDosageMap()
    rt.Shade()
    comp = orc.Computation(oscene, oroute["lamps"][:1], 400000, oroute["lightHeight"], oroute["lightLength"], power)
    comp.SEED = seed
    comp.reset()
    comp.iteration()
    assert np.array_equal(bits(rt.read_dosage()), bits(comp.dose()))
    rt.close()


def test_launch_sharding_union_equals_single(host, orc, oscene, oroute):
    """Two 'ranks' (two contexts on the one GPU) each run every other lamp launch; SUM / MAX of
    their maps equals the single-context computation exactly (DESIGN.md Multi-GPU)."""
    lamps = oroute["lamps"][:4]
    comp = orc.Computation(oscene, lamps, 200000, oroute["lightHeight"], oroute["lightLength"], oroute["lightIntensity"])
    comp.reset()
    comp.iteration()
    comp.iteration()
    maps = []
    for rank in range(2):
        rt = host.RayTracer(GLB, ROUTE, device=0)
        rt.set_lamps(rt.lamps()[:4])
        rt.photonCount = 200000
        rt.set_shard(rank, 2)
        rt.ResetDosageMap()
        rt.ComputeDosageMap()
        rt.ComputeDosageMap()
        rt.Sync()
        assert rt.photonMapSize == comp.photonMapSize
        assert rt.ctx.seed == comp.SEED
        maps.append((rt.ctx.read_photon_map(0), rt.ctx.read_photon_map(1)))
        rt.close()
    assert np.array_equal(maps[0][0] + maps[1][0], comp.photonMap)
    assert np.array_equal(np.maximum(maps[0][1], maps[1][1]), comp.maxPhotonMap)


def test_device_maps_alias_as_torch_tensors(host, pkg):
    """sharding.wrap_map: the context's f64 maps as zero-copy torch tensors (what the RCCL
    reduction operates on)."""
    torch = pytest.importorskip("torch")
    from uvrt_amd import sharding
    rt = host.RayTracer(GLB, ROUTE, device=0)
    rt.set_lamps(rt.lamps()[:1])
    rt.photonCount = 100000
    rt.ResetDosageMap()
    rt.ComputeDosageMap()
    rt.Sync()
    dev = torch.device("cuda", 0)
    t_sum = sharding.wrap_map(rt.ctx, 0, dev)
    t_max = sharding.wrap_map(rt.ctx, 1, dev)
    assert t_sum.dtype == torch.float64 and t_sum.numel() == rt.mesh.triangleCount
    assert np.array_equal(t_sum.cpu().numpy(), rt.ctx.read_photon_map(0))
    assert np.array_equal(t_max.cpu().numpy(), rt.ctx.read_photon_map(1))
    t_sum.mul_(2.0)                       # written through torch, seen by the context
    torch.cuda.synchronize()
    assert np.array_equal(rt.ctx.read_photon_map(0), t_sum.cpu().numpy())
    red = sharding.MapReducer(rt.ctx, dev)
    assert red.staged is False
    red()                                 # world size 1: a no-op
    rt.close()
