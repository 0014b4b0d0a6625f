"""Developer probe: extend time with the node-pair records renumbered so that the records lamp 0's
rays visit most (oracle visit counts over a sample) form the LDS-cached prefix (uvrt_set_record_perm)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g

pkg = g.load_package(); orc = g.load_oracle()
glb = os.path.join(ROOT, "tests/golden/testroomopt.glb")
route = orc.load_route(os.path.join(ROOT, "tests/golden/lange_route.xml"))
s = orc.Scene(glb)
comp = orc.Computation(s, route["lamps"], 1 << 16, route["lightHeight"], route["lightLength"], route["lightIntensity"])
lp = comp.lamp_world_pos(route["lamps"][0])
n = 2073600
sample = int(os.environ.get("SAMPLE", 65536))
rays, _ = orc.generate(0, sample, lp, route["lightLength"], 12345)     # a different seed than the timed launch
hist = orc.extend_visit_hist(s.tris, rays, s.nodes, s.triIdx)
nodes = s.nodes
q = [0] if nodes[0]["triCount"] == 0 else []
i = 0
while i < len(q):
    l = int(nodes[q[i]]["leftFirst"]); i += 1
    for k in (0, 1):
        if nodes[l + k]["triCount"] == 0:
            q.append(l + k)
visits = hist[np.array(q)].astype(np.int64)
order = np.argsort(-visits, kind="stable")          # hottest first
perm = np.empty(len(q), dtype=np.uint32)
perm[order] = np.arange(len(q), dtype=np.uint32)
print("coverage of the first 127: BFS %.3f, hot %.3f" % (visits[:127].sum() / visits.sum(), visits[order[:127]].sum() / visits.sum()))

c = pkg.capi.Ctx(0)
c.set_scene(s.tris, s.nodes, s.triIdx)
c.resize_rays(n)
full, _ = orc.generate(0, n, lp, route["lightLength"], 0)
ref = np.zeros(s.T, dtype=np.int32)
orc.extend(ref, s.tris, full, s.nodes, s.triIdx)
best = {}
for rnd in range(4):
    for name, pm in (("bfs", None), ("hot", perm)):
        c.set_record_perm(pm)
        c.set_timing(True); c.reset(False); c.seed = 0
        c.generate(lp, route["lightLength"], 0, n); c.extend(n); c.sync()
        ok = np.array_equal(c.read_counts(), ref)
        c.extend_time_ms()
        for _ in range(5):
            c.seed = 0
            c.generate(lp, route["lightLength"], 0, n); c.extend(n); c.accumulate(60.0)
        c.sync()
        ms, k = c.extend_time_ms()
        best[name] = min(best.get(name, 1e9), ms / k)
        if rnd == 0:
            print(name, "counts", "OK" if ok else "MISMATCH", flush=True)
for name, ms in best.items():
    print("%s: extend %.3f ms (%.1f Mray/s)" % (name, ms, n / ms / 1e3))
