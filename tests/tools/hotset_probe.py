"""Developer probe (not the bench): the hot-record set-up ALONE, once per lamp of the route -- every lamp position is new
to the context, each generate call is followed by a device sync, nothing else runs on the GPU.  Under
`rocprofv3 --kernel-trace --stats` this gives k_visit_stats / k_select_hot / k_write_perm per single lamp (the first call is
the process's cold one).  With QUALITY=1 (default) it prints the coverage of the selection against the ORACLE's visit counts
of the same sample rays, as tests/test_gpu_hotset.py asserts it.

    LAMPS=12 python3 tests/tools/hotset_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402

import torch  # noqa: E402

torch.cuda.init()
pkg = g.load_package()
orc = g.load_oracle()          # a developer tool: the checker of the selection's quality
from test_gpu_hotset import KEEP, SAMPLE, lamp_pos, pair_order  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
oscene = orc.Scene(os.path.join(GOLDEN, "testroomopt.glb"))
oroute = orc.load_route(os.path.join(GOLDEN, "lange_route.xml"))
nl = int(os.environ.get("LAMPS", "12"))
quality = os.environ.get("QUALITY", "1") == "1"
order = pair_order(oscene.nodes)
c = pkg.capi.Ctx(0)
c.set_scene(oscene.tris, oscene.nodes, oscene.triIdx)
n = 65536
c.resize_rays(n)
cov, share = [], []
for k in range(nl):
    lp = lamp_pos(orc, oscene, oroute, k % 12)
    lp = (lp[0] + 0.003 * (k // 12), lp[1], lp[2])
    c.seed = 17 * k
    c.generate(lp, oroute["lightLength"], 0, n)
    c.sync()
    if quality:
        perm = c.read_record_perm(order.size)
        hot = np.flatnonzero(perm < KEEP)
        rays, _ = orc.generate(0, SAMPLE, lp, oroute["lightLength"], 17 * k)
        visits = orc.extend_visit_hist(oscene.tris, rays, oscene.nodes, oscene.triIdx)[order].astype(np.int64)
        best = np.sort(visits)[::-1][:KEEP].sum()
        cov.append(visits[hot].sum() / best)
        share.append(visits[hot].sum() / visits.sum())
if quality:
    print("coverage of the best %d over %d lamps: min %.4f mean %.4f; share of all inner visits: mean %.4f" %
          (KEEP, nl, min(cov), sum(cov) / nl, sum(share) / nl), flush=True)
c.close()
