"""Developer probe (CPU): oracle/uvrt_oracle.c orc_wave_sim -- the persistent-wave scheduler of k_extend6 replayed with the
real arithmetic, as it is (mode 0) and with the deferred triangle queue of VERDICT r3 item 3 (mode 1).
    N=2073600 RPW=771 python tests/tools/wave_sim.py"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
orc = g.load_oracle()
L = orc.lib()
names = ["rays", "trips", "inner_trips", "leaf_blocks", "inner_lane_visits", "leaf_lane_tests", "wait_lane_trips", "idle_lane_trips",
         "refills", "flushes", "queue_entries", "differ_tri", "differ_dist", "ref_inner_visits", "ref_tri_tests"]
class Sim(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in names]
L.orc_wave_sim.argtypes = [C.c_void_p] * 2 + [C.c_int64] + [C.c_void_p] * 2 + [C.c_int] * 4 + [C.POINTER(Sim), C.c_int]
s = orc.Scene(os.path.join(ROOT, "tests/golden/testroomopt.glb"))
route = orc.load_route(os.path.join(ROOT, "tests/golden/lange_route.xml"))
comp = orc.Computation(s, route["lamps"], 1 << 16, route["lightHeight"], route["lightLength"], route["lightIntensity"])
n = int(os.environ.get("N", "524288"))
rpw = int(os.environ.get("RPW", "771"))
lamp = int(os.environ.get("LAMP", "0"))
rays, _ = orc.generate(0, n, comp.lamp_world_pos(route["lamps"][lamp]), route["lightLength"], 0)
def p(a): return a.ctypes.data_as(C.c_void_p)
def run(mode, refill, q):
    st = Sim()
    L.orc_wave_sim(p(s.tris), p(rays), n, p(s.nodes), p(s.triIdx), mode, refill, q, rpw, C.byref(st), 0)
    return {k: getattr(st, k) for k in names}
print("n %d rays, %d rays per wave, lamp %d" % (n, rpw, lamp))
base = None
for mode, refill, q in [(0, 8, 0), (1, 8, 32), (1, 8, 48), (1, 8, 64), (1, 16, 48), (1, 16, 64), (1, 24, 64)]:
    d = run(mode, refill, q)
    r = d["rays"]
    line = ("mode %d refill %2d flush %2d: trips/ray %.4f  inner-block trips/ray %.4f  tri blocks/ray %.4f  inner lane-visits/ray %.3f (ref %.3f)  "
            "leaf tests/ray %.3f (ref %.3f)  lanes: inner %.1f wait %.1f idle %.1f  entries/flush %.1f  differ: tri %d dist %d"
            % (mode, refill, q, d["trips"] / r, d["inner_trips"] / r, d["leaf_blocks"] / r, d["inner_lane_visits"] / r, d["ref_inner_visits"] / r,
               d["leaf_lane_tests"] / r, d["ref_tri_tests"] / r, d["inner_lane_visits"] / d["trips"], d["wait_lane_trips"] / d["trips"],
               d["idle_lane_trips"] / d["trips"], d["queue_entries"] / max(1, d["flushes"]), d["differ_tri"], d["differ_dist"]))
    print(line, flush=True)
