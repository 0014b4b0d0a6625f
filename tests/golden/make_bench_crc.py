"""Regenerates tests/golden/bench_dose_crc.json with the ORACLE (CPU restatement of the reference kernels):

    python tests/golden/make_bench_crc.py [--skip-route]

zlib.crc32 of the f32 dose of the workloads bench.py times, so that every bench run -- also at N > 1, where the
oracle leg does not run, and for the modes the in-run oracle leg does not cover -- can check the dose of its timed
region against a committed value:

  flavour0 / flavour1     8 waves x 2 073 600 photons, lamp 0 of lange_route.xml, SEED_0 = 0 (bench.py default)
  seed1_flavour1          the same workload under `uvrt_set_seed_mode(1)` + `uvrt_set_flavour(1)`: what the reference's
                          own kernels do on gfx950 (every work-item reads SEED_{k-1}; a negative seed sum converts to 0;
                          fused cross/dot in IntersectTri)
  route_flavour0          the reference's DEFAULT workload (raytracer.h:30-32, positions/lange_route.xml,
                          myapp.cpp:156-170): 12 lamps x 10 iterations x ((2^25 / 12) & ~1) photons
"""
import json
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

PHOTONS, WAVES = 1920 * 1080, 8
ROUTE_PHOTONS, ROUTE_ITERATIONS = 1 << 25, 10      # raytracer.h:30-32


def crc(a):
    return "%08x" % zlib.crc32(np.ascontiguousarray(a).tobytes())


def main():
    threads = len(os.sched_getaffinity(0))
    s = orc.Scene(os.path.join(HERE, "testroomopt.glb"))
    r = orc.load_route(os.path.join(HERE, "lange_route.xml"))
    path = os.path.join(HERE, "bench_dose_crc.json")
    out = json.load(open(path)) if os.path.exists(path) else {}

    def lamp0(flavour, seed_mode):
        orc.set_flavour(flavour)
        c = orc.Computation(s, r["lamps"][:1], PHOTONS, r["lightHeight"], r["lightLength"], r["lightIntensity"], nthreads=threads)
        c.reset()
        lamp = r["lamps"][0]
        lp = c.lamp_world_pos(lamp)
        for _ in range(WAVES):
            if seed_mode == 0:
                rays, c.SEED = orc.generate(0, PHOTONS, lp, c.lightLength, c.SEED)
            else:
                rays, c.SEED = orc.generate_fixed_seed(0, PHOTONS, lp, c.lightLength, c.SEED, saturate=True)
            orc.extend(c.temp, s.tris, rays, s.nodes, s.triIdx, threads)
            orc.accumulate(c.photonMap, c.maxPhotonMap, c.temp, lamp[2])
            c.photonMapSize += PHOTONS
        orc.set_flavour(0)
        return crc(c.dose())

    t0 = time.time()
    out["flavour0"] = lamp0(0, 0)
    out["flavour1"] = lamp0(1, 0)
    out["seed1_flavour1"] = lamp0(1, 1)
    print("lamp-0 workloads: %.1f s" % (time.time() - t0), out, flush=True)
    if "--skip-route" not in sys.argv:
        t0 = time.time()
        c = orc.Computation(s, r["lamps"], ROUTE_PHOTONS, r["lightHeight"], r["lightLength"], r["lightIntensity"], nthreads=threads)
        c.reset()
        for it in range(ROUTE_ITERATIONS):
            c.iteration()
            print("route iteration %d: %.1f s" % (it + 1, time.time() - t0), flush=True)
        out["route_flavour0"] = crc(c.dose())
        out["route_workload"] = {"lamps": len(r["lamps"]), "iterations": ROUTE_ITERATIONS, "photons_per_lamp_launch": c.photonsPerLight,
                                 "rays": c.photonMapSize, "final_SEED": "%#x" % c.SEED,
                                 "dose_sum": float(c.dose().astype(np.float64).sum())}
    out["what"] = ("zlib.crc32 of the f32 dose computed by the ORACLE (tests/golden/make_bench_crc.py): flavour0 / flavour1 / "
                   "seed1_flavour1 = 8 waves x 2 073 600 photons, lamp 0 of lange_route.xml, testroomopt.glb, SEED_0 = 0 (bench.py "
                   "default config) in the canonical arithmetic, with the fused cross/dot of ROCm's OpenCL library, and with that "
                   "plus the SEED semantics the reference's generate.cl has on gfx950; route_flavour0 = the reference's default "
                   "workload, 12 lamps x 10 iterations x 2 796 202 photons (bench.py --route)")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
