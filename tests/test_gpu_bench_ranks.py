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
    two = bench(common + ["--waves", "6"], ranks=2, port=29531)        # strong headline; weak: 2 ranks x 6 waves
    three = bench(common + ["--waves", "6", "--scaling", "weak"], ranks=3, port=29532)
    assert one["other_modes"]["loop"]["dose_crc32"] == one["dose_crc32"] and one["config"]["mode"] == "batched"
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and three["scaling"] == "weak"
    for d in (two, three):
        assert d["multi_gpu_check"]["dose_identical_on_all_ranks"]
        # ray-range shards + ONE reduction of the count planes per computation = the single-GPU dose bits
        assert d["strong"]["dose_crc32"] == one["dose_crc32"]
        assert d["strong"]["rays_per_step"] == one["config"]["rays_per_step"]
    assert two["dose_crc32"] == one["dose_crc32"] and two["value"] == two["strong"]["value"]
    assert three["value"] == three["weak"]["value"] and three["weak"]["rays_per_step"] == 3 * one["config"]["rays_per_step"]
    # weak at 2 ranks traces 12 launches of the same SEED chain: its dose is the 12-wave single-GPU dose
    twelve = bench(common + ["--waves", "12"])
    assert two["weak"]["dose_crc32"] == twelve["dose_crc32"]
    for d in (one, two, three):
        assert d["roofline"] is None or 0.0 < d["roofline"]["frac"] <= 1.0
        assert d["unit"] == "Mray/s" and d["higher_is_better"] is True


def test_bench_starts_its_own_ranks_and_labels_the_rehearsal():
    """`python3 bench.py --gpus 2 --steps 5` -- no launcher, no WORLD_SIZE -- must produce a line by
    itself (the parent touches no GPU and starts one child per rank); on a 1-GPU box that is the gloo REHEARSAL, whose
    dose CRC equals the N = 1 CRC."""
    common = ["--photons", "200000", "--warmup", "1", "--no-cpu-baseline", "--waves", "4"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5"] + common,
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                  # rank 0's line only
    two = json.loads(lines[0])
    one = bench(common + ["--steps", "1", "--lean"])
    assert two["n_gpus"] == 2 and two["steps"] == 5 and two["comm"].startswith("REHEARSAL")
    assert two["dose_crc32"] == one["dose_crc32"] and two["ranks_agree"] is True
    assert two["strong"]["dose_crc32"] == one["dose_crc32"] and two["weak"] is not None
    assert one["comm"] is None and one["rccl_ranks"] is None and one["reserved_cus"] == 0


def test_single_rank_rehearsals_of_both_collective_branches():
    """The N > 1 step with the context's own RCCL communicator (uvrt_reduce_batch) and with torch.distributed's RCCL
    all-reduce of the device planes (the fallback branch) -- each run for real with the one rank a 1-GPU box allows:
    same dose bits as the plain step, and RCCL itself reports the communicator's span."""
    common = ["--photons", "200000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--waves", "4", "--lean"]
    plain = bench(common)
    native = bench(common + ["--self-comm"])
    fallback = bench(common + ["--self-comm", "--comm", "torch"])
    assert native["comm"] == "native" and native["rccl_ranks"] == 1 and native["reserved_cus"] in (0, 8)
    assert fallback["comm"] == "torch" and fallback["rccl_ranks"] == 1 and fallback["reserved_cus"] == 0
    assert native["dose_crc32"] == plain["dose_crc32"] == fallback["dose_crc32"]
