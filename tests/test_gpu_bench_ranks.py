"""GPU: bench.py's multi-rank paths, rehearsed with two ranks sharing the one GPU (collective over
gloo): the launch-sharded (weak) and the ray-range-sharded (strong) computation must produce the
same dose bits as one rank tracing the same launches."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def bench(args, ranks=1, port=29530):
    env = dict(os.environ)
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("LOCAL_RANK", None)
    if ranks == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
               "--gpus", str(ranks)] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def test_two_rank_rehearsal_matches_single_rank():
    common = ["--photons", "300000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
    one = bench(common + ["--waves", "6"])
    weak = bench(common + ["--waves", "3"], ranks=2, port=29531)                 # 2 ranks x 3 waves = 6 launches
    strong = bench(common + ["--waves", "6", "--scaling", "strong"], ranks=2, port=29532)
    assert weak["n_gpus"] == strong["n_gpus"] == 2 and weak["scaling"] == "weak" and strong["scaling"] == "strong"
    assert weak["multi_gpu_check"]["dose_identical_on_all_ranks"]
    assert strong["multi_gpu_check"]["dose_identical_on_all_ranks"]
    assert weak["config"]["rays_per_step"] == strong["config"]["rays_per_step"] == one["config"]["rays_per_step"]
    assert weak["dose_crc32"] == one["dose_crc32"]
    assert strong["dose_crc32"] == one["dose_crc32"]
    for d in (one, weak, strong):
        assert d["roofline"] is None or 0.0 < d["roofline"]["frac"] <= 1.0
        assert d["unit"] == "Mray/s" and d["higher_is_better"] is True
