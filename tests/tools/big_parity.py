"""Developer probe: one large bit-exactness run of the default path -- every lamp of the route,
2 073 600 photons each, per-ray (dist, triID) and counts against the oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
pkg = g.load_package(); orc = g.load_oracle()
s = orc.Scene(os.path.join(ROOT, "tests/golden/testroomopt.glb"))
route = orc.load_route(os.path.join(ROOT, "tests/golden/lange_route.xml"))
comp = orc.Computation(s, route["lamps"], 1 << 16, route["lightHeight"], route["lightLength"], route["lightIntensity"])
n = 2073600
c = pkg.capi.Ctx(0)
c.set_scene(s.tris, s.nodes, s.triIdx)
c.resize_rays(n)
c.set_record_hits(True)
seed = 0
tot = 0
for li, lamp in enumerate(route["lamps"]):
    lp = comp.lamp_world_pos(lamp)
    rays, nseed = orc.generate(0, n, lp, route["lightLength"], seed)
    temp = np.zeros(s.T, dtype=np.int32)
    orc.extend(temp, s.tris, rays, s.nodes, s.triIdx)
    c.reset(False); c.seed = seed
    c.generate(lp, route["lightLength"], 0, n); c.extend(n); c.sync()
    got = c.read_rays(0, n)
    ok = (np.array_equal(got["dist"].view(np.uint32), rays["dist"].view(np.uint32)) and
          np.array_equal(got["triID"], rays["triID"]) and np.array_equal(c.read_counts(), temp) and c.seed == nseed)
    print("lamp %2d: %s  hits %d" % (li, "bit-identical" if ok else "MISMATCH", int(temp.sum())), flush=True)
    assert ok
    seed = nseed; tot += n
print("all %d rays bit-identical" % tot)
