"""Developer probe: one large bit-exactness run of the default path -- every lamp of the route,
2 073 600 photons each, per-ray (dist, triID) and counts against the oracle.
FLAVOUR=2: the "shipped flags" arithmetic against the oracle in flavour 2 AND against the reference's own extend.cl built
with its own flags, running on this GPU (oracle/_ref/ref_extend_fast.co); also counts the rays that differ from flavour 0."""
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
FL = int(os.environ.get("FLAVOUR", "0"))
orc.set_flavour(FL)
c = pkg.capi.Ctx(0)
c.set_flavour(FL)
diff_tri = diff_bits = 0
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
    extra = ""
    if FL == 2:
        rr = rays.copy(); rr["dist"] = np.float32(1e30); rr["triID"] = 0
        rc, _ = orc.refgpu_extend(rr, s.tris, s.nodes, s.triIdx, shipped=True)
        ok = ok and (np.array_equal(got["dist"].view(np.uint32), rr["dist"].view(np.uint32)) and
                     np.array_equal(got["triID"], rr["triID"]) and np.array_equal(rc, temp))
        orc.set_flavour(0)
        r0 = rays.copy(); r0["dist"] = np.float32(1e30); r0["triID"] = 0
        t0 = np.zeros(s.T, dtype=np.int32)
        orc.extend(t0, s.tris, r0, s.nodes, s.triIdx)
        orc.set_flavour(2)
        dt = int((r0["triID"] != rays["triID"]).sum()); db = int((r0["dist"].view(np.uint32) != rays["dist"].view(np.uint32)).sum())
        diff_tri += dt; diff_bits += db
        extra = "  = reference kernel built with its own flags; vs flavour 0: %d rays on another triangle, %d with other dist bits" % (dt, db)
    print("lamp %2d: %s  hits %d%s" % (li, "bit-identical" if ok else "MISMATCH", int(temp.sum()), extra), flush=True)
    assert ok
    seed = nseed; tot += n
print("all %d rays bit-identical (flavour %d)" % (tot, FL))
if FL == 2:
    print("against the strict flavour 0: %d of %d rays hit another triangle (%.2e), %d differ in dist bits (%.3f)"
          % (diff_tri, tot, diff_tri / tot, diff_bits, diff_bits / tot))
