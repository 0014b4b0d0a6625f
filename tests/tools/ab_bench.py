"""Developer probe (not the bench): interleaved A/B of extend-kernel knob settings (uvrt_set_variant) on the bench's own
step, all in ONE process -- rounds x variants, each cell = STEPS back-to-back computations bracketed by device syncs
(cdna_hip_programming.md rule 24: perf deltas come from interleaved rounds in one process).

    VARIANTS=0,1208,1308 MODE=batched STEPS=20 ROUNDS=5 python tests/tools/ab_bench.py

Prints per variant the median and best Mray/s over the rounds, the dose CRC (must be the same for every variant) and,
with ISOLATED=1, the HIP-event duration of an extend launch on one stream (launch pipelining off)."""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

import torch  # noqa: E402  (HIP runtime order: torch first, see tests/conftest.py)

torch.cuda.init()
g.load_package()
from uvrt_amd import host  # noqa: E402

variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
# FLAVOURS=0,2: every variant in each of these arithmetic flavours (uvrt_set_flavour), interleaved like the variants; a
# cell is then named 10000 * flavour + variant.  The dose CRC must agree within a flavour.
flavours = [int(v) for v in os.environ.get("FLAVOURS", os.environ.get("FLAVOUR", "0")).split(",")]
# WIDES=0,1: the default BVH2 walk / the opt-in 4-wide walk (uvrt_set_wide_bvh; loop modes only -- batches use the default walk)
wides = [int(v) for v in os.environ.get("WIDES", os.environ.get("WIDE", "0")).split(",")]
cells = [(f, v, w) for f in flavours for v in variants for w in wides]
mode = os.environ.get("MODE", "batched")
steps = int(os.environ.get("STEPS", "20"))
rounds = int(os.environ.get("ROUNDS", "5"))
photons = int(os.environ.get("PHOTONS", str(1920 * 1080)))
waves = int(os.environ.get("WAVES", "8"))
nlamps = int(os.environ.get("LAMPS", "1"))
scene = os.environ.get("SCENE", "")

glb = os.path.join(ROOT, "tests", "golden", "testroomopt.glb")
route_xml = os.path.join(ROOT, "tests", "golden", "lange_route.xml")
if scene.startswith("soup:"):
    sys.path.insert(0, ROOT)
    import bench
    mesh = host.Mesh(tris=bench.soup_triangles(int(scene[5:])))
    rt = host.RayTracer(None, route_xml, device=0, mesh=mesh)
else:
    rt = host.RayTracer(glb, route_xml, device=0)
rt.set_lamps(rt.lamps()[:nlamps])
rt.photonCount = photons * nlamps
rt.maxIterations = waves
rays_per_step = waves * rt.photonsPerLight * nlamps


def step():
    rt.ctx.seed = 0
    rt.ResetDosageMap()
    if mode == "batched":
        rt.ComputeIterationsBatched(waves)
    else:
        for _ in range(waves):
            rt.ComputeDosageMap()
            rt.Shade()
            rt.currIterations = rt.currIterations + 1
            if mode == "loop_sync":
                rt.Sync()


res = {c: [] for c in cells}
crcs = {}
iso = {}
for rnd in range(rounds + 1):                 # round 0 = warm-up (allocations, hot records, clocks)
    for c in cells:
        f, v, w = c
        rt.ctx.set_flavour(f)
        rt.ctx.set_variant(v)
        rt.ctx.set_wide_bvh(bool(w))
        step()
        rt.Sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        rt.Sync()
        el = time.perf_counter() - t0
        if rnd > 0:
            res[c].append(rays_per_step * steps / el / 1e6)
        crcs[c] = "%08x" % zlib.crc32(rt.read_dosage().tobytes())
if os.environ.get("ISOLATED", "0") == "1":
    rt.ctx.set_pipeline(False)
    for rnd in range(3):
        for c in cells:
            f, v, w = c
            rt.ctx.set_flavour(f)
            rt.ctx.set_variant(v)
            rt.ctx.set_wide_bvh(bool(w))
            rt.ctx.set_timing(True)
            rt.ctx.extend_time_ms()
            rt.ctx.seed = 0
            rt.ResetDosageMap()
            for _ in range(waves):
                rt.ComputeDosageMap()
            rt.Sync()
            ms, k = rt.ctx.extend_time_ms()
            rt.ctx.set_timing(False)
            iso.setdefault(c, []).append(ms / max(k, 1))
    rt.ctx.set_pipeline(True)
base = np.median(res[cells[0]])
for c in cells:
    a = np.array(res[c])
    line = "variant %5d flavour %d wide %d  %s  median %8.1f  best %8.1f  min %8.1f Mray/s  (%+.2f %% vs the first)  crc %s" % (
        c[1], c[0], c[2], mode, np.median(a), a.max(), a.min(), 100.0 * (np.median(a) / base - 1.0), crcs[c])
    if c in iso:
        line += "  isolated extend %.4f ms" % min(iso[c])
    print(line, flush=True)
for f in flavours:
    if len({crcs[c] for c in cells if c[0] == f}) != 1:
        print("DOSE MISMATCH between variants of flavour", f, crcs)
        sys.exit(1)
