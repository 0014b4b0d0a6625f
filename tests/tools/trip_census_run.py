"""Developer probe (GPU box, library built with -DUVRT_TRIP_STATS): K computations of the bench's step in MODE (batched / loop),
one sync at the end -- the library prints its 'trip census:' line there -- and the number of rays they traced."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
import torch
torch.cuda.init()
g.load_package()
from uvrt_amd import host
mode = os.environ.get("MODE", "batched")
k = int(os.environ.get("COMPUTATIONS", "2"))
waves = 8
rt = host.RayTracer(os.path.join(ROOT, "tests/golden/testroomopt.glb"), os.path.join(ROOT, "tests/golden/lange_route.xml"), device=0)
rt.set_lamps(rt.lamps()[:1])
rt.photonCount = 1920 * 1080
rt.maxIterations = waves
rt.ctx.set_flavour(int(os.environ.get("FLAVOUR", "0")))
if mode != "batched" and os.environ.get("PIPELINE", "1") == "0":
    rt.ctx.set_pipeline(False)
rt.Sync()          # (prints and clears whatever the set-up traced: nothing)
for _ in range(k):
    rt.ctx.seed = 0
    rt.ResetDosageMap()
    if mode == "batched":
        rt.ComputeIterationsBatched(waves)
    else:
        for _ in range(waves):
            rt.ComputeDosageMap()
            rt.Shade()
            rt.currIterations = rt.currIterations + 1
rt.Sync()
sys.stderr.write("census rays %d\n" % (k * waves * rt.photonsPerLight))
