"""Developer probe: do consecutive launches overlap when they alternate between two streams?
Two contexts on one GPU (each its own stream and buffers), launches dealt alternately, against one
context running the same number of launches back to back."""
import os, sys, time
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
ctxs = [pkg.capi.Ctx(0) for _ in range(3)]
for c in ctxs:
    c.set_scene(s.tris, s.nodes, s.triIdx); c.resize_rays(n); c.reset(False)
    c.set_variant(int(os.environ.get("VARIANT", "0")))
    c.set_pipeline(os.environ.get("PIPE", "0") == "1")      # PIPE=1: every context also pipelines over its two lanes


def run(cs, launches=32):
    for c in cs: c.sync()
    t0 = time.time()
    for k in range(launches):
        c = cs[k % len(cs)]
        c.generate(lp, route["lightLength"], 0, n); c.extend(n); c.accumulate(60.0)
        c.shade(0, n, 44.0, 100.0, False)
    for c in cs: c.sync()
    return (time.time() - t0) / launches * 1e3


for rnd in range(3):
    print("one context: %.3f ms per launch   two contexts: %.3f   three contexts: %.3f"
          % (run(ctxs[:1]), run(ctxs[:2]), run(ctxs[:3])), flush=True)
