"""Regenerates the oracle-made fixtures in this directory (run from the repo root):

    python tests/golden/make_golden.py

  oracle_small.npz              lamp 0, one launch of 65 536 photons, SEED_0 = 0:
                                rays[0:256] (32-byte records), (dist, triID) of rays[0:4096],
                                the full int32 count vector, dose[0:2048] and the dose checksum
  census_lamp0_2073600.json     traversal census of the bench workload (8 launches of 2 073 600
                                photons, lamp 0, SEED chain from 0): prices the algorithmic bytes
                                per ray of SURVEY.md 8d

The oracle is pinned on SURVEY.md 8c (tests/test_oracle_golden.py); these files make that
pin available to the GPU tests without re-running the CPU code at full size.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402


def main():
    s = orc.Scene(os.path.join(HERE, "testroomopt.glb"))
    r = orc.load_route(os.path.join(HERE, "lange_route.xml"))
    c = orc.Computation(s, r["lamps"][:1], 65536, r["lightHeight"], r["lightLength"], r["lightIntensity"])
    c.reset()
    lp = c.lamp_world_pos(r["lamps"][0])
    rays, seed1 = orc.generate(0, 65536, lp, r["lightLength"], 0)
    temp = np.zeros(s.T, dtype=np.int32)
    orc.extend(temp, s.tris, rays, s.nodes, s.triIdx)
    pm, mm = np.zeros(s.T), np.zeros(s.T)
    counts = temp.copy()
    orc.accumulate(pm, mm, temp, r["lamps"][0][2])
    dose = orc.compute_dosage(pm, s.tris, 65536, np.float32(np.float32(r["lightIntensity"]) * np.float32(0.1)))
    np.savez_compressed(os.path.join(HERE, "oracle_small.npz"),
                        light_pos=np.array(lp, dtype=np.float32), seed1=np.uint32(seed1),
                        rays256=rays[:256].copy(), dist4096=rays["dist"][:4096].copy(),
                        tri4096=rays["triID"][:4096].copy(), counts=counts, dose2048=dose[:2048].copy(),
                        dose_sum=np.float64(dose.astype(np.float64).sum()))
    # bench workload census
    c = orc.Computation(s, r["lamps"][:1], 2073600, r["lightHeight"], r["lightLength"], r["lightIntensity"])
    c.reset()
    for _ in range(8):
        c.iteration()
    keys = ("rays", "node_visits", "aabb_tests", "tri_tests", "hits")
    avg = {k: sum(st[k] for st in c.stats) / len(c.stats) for k in keys}
    out = {"workload": "testroomopt.glb, lamp 0 of lange_route.xml, 8 launches x 2073600 photons, SEED_0=0",
           "per_launch": c.stats, "per_launch_avg": avg,
           "algorithmic_bytes_per_ray": orc.algorithmic_bytes_per_ray(avg),
           "max_stack": max(st["max_stack"] for st in c.stats), "final_SEED": "%#x" % c.SEED,
           "dose_sum": float(c.dose().astype(np.float64).sum())}
    json.dump(out, open(os.path.join(HERE, "census_lamp0_2073600.json"), "w"), indent=1)
    print(json.dumps(out["per_launch_avg"]), out["algorithmic_bytes_per_ray"])


if __name__ == "__main__":
    main()
