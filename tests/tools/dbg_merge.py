import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
s = orc.Scene(os.path.join(ROOT, "tests/golden/testroomopt.glb"))
route = orc.load_route(os.path.join(ROOT, "tests/golden/lange_route.xml"))
comp = orc.Computation(s, route["lamps"], 1 << 16, route["lightHeight"], route["lightLength"], route["lightIntensity"])
lp = comp.lamp_world_pos(route["lamps"][3])
c = pkg.capi.Ctx(0)
c.set_scene(s.tris, s.nodes, s.triIdx)
for n in (200000, 2073600):
  for rec in (True, False):
    c.set_record_hits(rec)
    c.resize_rays(n); c.reset(False); c.seed = 0
    c.generate(lp, route["lightLength"], 0, n); c.extend(n); c.sync()
    got = c.read_rays(0, n); counts = c.read_counts()
    rays, _ = orc.generate(0, n, lp, route["lightLength"], 0)
    temp = np.zeros(s.T, dtype=np.int32)
    orc.extend(temp, s.tris, rays, s.nodes, s.triIdx)
    bd = np.flatnonzero(got["dist"].view(np.uint32) != rays["dist"].view(np.uint32)) if rec else np.zeros(0, int)
    bt = np.flatnonzero(got["triID"] != rays["triID"]) if rec else np.zeros(0, int)
    print("n", n, "record", rec, "dist mismatches", bd.size, "tri mismatches", bt.size, "count diffs", int((counts != temp).sum()),
          "sum got", int(counts.sum()), "want", int(temp.sum()))
    if bd.size:
        print("  first bad", bd[:10], "got", got["dist"][bd[:6]], "want", rays["dist"][bd[:6]], "got tri", got["triID"][bd[:6]], "want", rays["triID"][bd[:6]])
        print("  bad % 64:", np.bincount(bd % 64, minlength=64).tolist())
