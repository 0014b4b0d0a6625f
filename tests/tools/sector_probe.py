"""Developer probe (GPU box): would an XCD-aware partition of the rays pay?  Upper bound: trace 2 073 600 rays that ALL lie in one
azimuth sector of the lamp (every XCD then touches only that sector's records: a working set of a few hundred KB instead of
5.7 MB) against 2 073 600 rays of all directions, in ns per traversal step (the oracle counts the steps of both sets)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
s = orc.Scene(os.path.join(ROOT, "tests/golden/testroomopt.glb"))
route = orc.load_route(os.path.join(ROOT, "tests/golden/lange_route.xml"))
comp = orc.Computation(s, route["lamps"], 1 << 16, route["lightHeight"], route["lightLength"], route["lightIntensity"])
lp = comp.lamp_world_pos(route["lamps"][0])
n = 2073600
sectors = int(os.environ.get("SECTORS", "8"))
big, _ = orc.generate(0, n * (sectors + 2), lp, route["lightLength"], 0)
phi = np.arctan2(big["dirz"], big["dirx"])
sets = {"all directions": big[:n]}
for k in (0, 3, 5):
    m = (phi >= -np.pi + 2 * np.pi * k / sectors) & (phi < -np.pi + 2 * np.pi * (k + 1) / sectors)
    sel = big[m][:n]
    if sel.size == n:
        sets["sector %d of %d" % (k, sectors)] = sel
steps = {k: float(orc.extend_steps(s.tris, v, s.nodes, s.triIdx).astype(np.float64).mean()) for k, v in sets.items()}
c = pkg.capi.Ctx(0)
c.set_scene(s.tris, s.nodes, s.triIdx)
c.resize_rays(n)
c.set_pipeline(False)
for fl in (0, 2):
    c.set_flavour(fl)
    for name, rays in sets.items():
        best = 1e9
        for rep in range(4):
            c.reset(False)
            c.write_rays(rays)
            c.set_timing(True); c.extend_time_ms()
            c.extend(n); c.sync()
            ms, k = c.extend_time_ms()
            c.set_timing(False)
            best = min(best, ms / max(k, 1))
        print("flavour %d  %-18s steps/ray %.2f  extend %.4f ms  = %.3f ns per step per 1e0 rays (%.1f G steps/s)"
              % (fl, name, steps[name], best, best * 1e6 / (n * steps[name]), n * steps[name] / best / 1e6), flush=True)
