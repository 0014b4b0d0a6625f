import os, sys, time
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
import torch; torch.cuda.init()
g.load_package()
from uvrt_amd import host
ROOT=os.getcwd()
rt = host.RayTracer(os.path.join(ROOT, "tests/golden/testroomopt.glb"), os.path.join(ROOT, "tests/golden/lange_route.xml"), device=0)
rt.set_lamps(rt.lamps()[:1]); rt.photonCount = 1920*1080; rt.maxIterations = 8
def step():
    rt.ctx.seed = 0; rt.ResetDosageMap(); rt.ComputeIterationsBatched(8)
rt.ctx.clock_probe_start(2000); print("idle clock MHz", rt.ctx.clock_probe_read())
for fl in (0, 2):
    rt.ctx.set_flavour(fl)
    for _ in range(5): step()
    rt.Sync()
    rt.ctx.clock_probe_start(8000)
    t0=time.perf_counter()
    for _ in range(10): step()
    rt.Sync(); el=time.perf_counter()-t0
    print("flavour", fl, "under load: clock MHz %.1f" % rt.ctx.clock_probe_read(), "step ms %.4f" % (el/10*1e3))
