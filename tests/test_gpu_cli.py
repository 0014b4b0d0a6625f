"""GPU: the headless driver (uvrt_cli = MyApp::Tick's compute block, myapp.cpp:156-175) end to
end: progress lines, raw / NPY dose dumps and the PLY heat map, against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GLB, GOLDEN, ROOT

pytestmark = pytest.mark.gpu

CLI = os.path.join(ROOT, "small-project-uv-robot-ray-tracer_amd", "uvrt_cli")


def test_cli_tick_loop_and_exports(tmp_path, orc, oscene, oroute):
    npy, raw, ply = tmp_path / "dose.npy", tmp_path / "dose.f32", tmp_path / "heat.ply"
    cmd = [CLI, "--room", GLB, "--route-dir", GOLDEN, "--route", "lange_route", "--lamps", "2", "--photons", "200000",
           "--iterations", "3", "--dump", str(npy), "--ply", str(ply)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.count("Progress: ") == 3 and "photon count: 600000" in out.stdout
    subprocess.run(cmd[:-4] + ["--dump", str(raw)], check=True, capture_output=True, timeout=300)
    dose = np.load(npy)
    assert np.array_equal(dose.view(np.uint32), np.fromfile(raw, dtype="<f4").view(np.uint32))
    comp = orc.Computation(oscene, oroute["lamps"][:2], 200000, oroute["lightHeight"], oroute["lightLength"],
                           oroute["lightIntensity"])
    comp.reset()
    for _ in range(3):
        comp.iteration()
    ref = comp.dose()
    assert np.array_equal(dose.view(np.uint32), ref.view(np.uint32))
    # PLY: header + 3 coloured vertices and one face per triangle
    b = ply.read_bytes()
    head, body = b.split(b"end_header\n", 1)
    T = oscene.T
    assert b"element vertex %d" % (3 * T) in head and b"element face %d" % T in head
    assert len(body) == 3 * T * 15 + T * 13
    v = np.frombuffer(body[:3 * T * 15], dtype=np.dtype([("p", "<f4", 3), ("c", "u1", 3)]))
    assert np.array_equal(v["p"].reshape(T, 9), oscene.tris[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]])
    col = orc.dosage_to_color(ref, oroute["minDosage"], False)
    assert np.array_equal(v["c"].reshape(T, 9), np.floor(np.clip(col, 0, 1) * np.float32(255.0) + np.float32(0.5)).astype(np.uint8))


def test_cli_sharded_and_batched_runs_give_the_same_dose(tmp_path):
    """uvrt_cli --gpus 3 (one process, three contexts on the one device, every launch split by global-id
    range, one sum of the count planes per batch, native host loop in C++) and --batch 2 against the plain
    per-iteration run: identical dose files."""
    base = [CLI, "--room", GLB, "--route-dir", GOLDEN, "--route", "lange_route", "--lamps", "3", "--photons", "150000",
            "--iterations", "4"]
    outs = {}
    for tag, extra in (("plain", []), ("batch", ["--batch", "2"]), ("gpus", ["--gpus", "3"]), ("gpus_batch", ["--gpus", "2", "--batch", "4"])):
        f = tmp_path / (tag + ".f32")
        out = subprocess.run(base + extra + ["--dump", str(f)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr + out.stdout
        outs[tag] = (np.fromfile(f, dtype="<u4"), out.stdout)
    for tag in ("batch", "gpus", "gpus_batch"):
        assert np.array_equal(outs[tag][0], outs["plain"][0]), tag
    assert outs["plain"][1].count("Progress: ") == 4 and outs["batch"][1].count("Progress: ") == 2
    assert "Sharding every launch over 3 contexts" in outs["gpus"][1] and outs["plain"][0].any()


def test_cli_flavour_option(tmp_path, orc, oscene, oroute):
    """uvrt_cli --flavour 2: the headless Tick loop in the arithmetic of the reference's own build flags (uvrt_set_flavour 2);
    the dose equals the oracle's in that flavour (v_rcp_f32 through the table read from this GPU) bit for bit, --flavour 1 equals
    the oracle in flavour 1, and a flavour that does not exist is refused."""
    base = [CLI, "--room", GLB, "--route-dir", GOLDEN, "--route", "lange_route", "--lamps", "2", "--photons", "200000",
            "--iterations", "2"]
    flavours = [1] + ([2] if orc.refgpu() is not None else [])
    for fl in flavours:
        f = tmp_path / ("dose_f%d.f32" % fl)
        out = subprocess.run(base + ["--flavour", str(fl), "--dump", str(f)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        orc.set_flavour(fl)
        try:
            comp = orc.Computation(oscene, oroute["lamps"][:2], 200000, oroute["lightHeight"], oroute["lightLength"],
                                   oroute["lightIntensity"])
            comp.reset()
            comp.iteration(); comp.iteration()
            ref = comp.dose()
        finally:
            orc.set_flavour(0)
        assert np.array_equal(np.fromfile(f, dtype="<u4"), ref.view(np.uint32)), fl
    out = subprocess.run(base + ["--flavour", "7"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 2 and "0, 1 or 2" in out.stderr
