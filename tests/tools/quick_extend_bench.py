"""Developer probe (not the bench): time extend for a few knob settings on lamp 0, checking
counts against the oracle once.  Lives under tests/ because it uses the oracle's scene loader."""
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
n = int(os.environ.get("N", 2073600))
variants = [int(v) for v in os.environ.get("VARIANTS", "0").split(",")]
sorts = [int(v) for v in os.environ.get("SORTS", "0,-1,12,15,18").split(",")]
c = pkg.capi.Ctx(0, dev=any(pkg.capi.needs_dev(v) for v in variants))
c.set_scene(s.tris, s.nodes, s.triIdx)
c.resize_rays(n)
c.set_flavour(int(os.environ.get("FLAVOUR", "0")))
if os.environ.get("WIDE", "0") == "1":
    c.set_wide_bvh(True)
if os.environ.get("HOT", "1") == "0":
    c.set_hot_records(0)
if os.environ.get("PIPELINE", "1") == "0":
    c.set_pipeline(False)
ref = None
if os.environ.get("CHECK", "1") == "1":
    rays, _ = orc.generate(0, n, lp, route["lightLength"], 0)
    ref = np.zeros(s.T, dtype=np.int32)
    st = orc.extend(ref, s.tris, rays, s.nodes, s.triIdx)
    print("oracle stats", st, "B/ray %.1f" % orc.algorithmic_bytes_per_ray(st), flush=True)
rounds = int(os.environ.get("ROUNDS", "3"))
best = {}
checked = {}
for rnd in range(rounds):          # interleave the configurations: clocks ramp up over the first runs
    for v in variants:
        for sb in sorts:
            c.set_variant(v); c.set_sort_bits(sb); c.set_timing(True)
            c.reset(False); c.seed = 0
            c.generate(lp, route["lightLength"], 0, n); c.extend(n); c.sync()
            if ref is not None and (v, sb) not in checked:
                checked[(v, sb)] = "OK" if np.array_equal(c.read_counts(), ref) else "MISMATCH"
            c.extend_time_ms()
            reps = 5
            c.sync(); t0 = time.time()
            for _ in range(reps):
                c.seed = 0
                c.generate(lp, route["lightLength"], 0, n); c.extend(n); c.accumulate(60.0)
            c.sync(); wall = (time.time() - t0) / reps
            ms, k = c.extend_time_ms()
            b = best.setdefault((v, sb), [1e9, 1e9])
            b[0] = min(b[0], ms / k); b[1] = min(b[1], wall * 1e3)
for (v, sb), (ms, wall) in best.items():
    print("variant %d sort_bits %3d: counts %s  extend %.3f ms  (%.1f Mray/s)  wave wall %.3f ms (%.1f Mray/s)"
          % (v, sb, checked.get((v, sb), "-"), ms, n / ms / 1e3, wall, n / wall / 1e3), flush=True)
