"""CPU: the kept bench lines are reproducible from the files under profiles/ -- every fraction of `roofline` follows from
the per-ray PMC model (profiles/extend_issue_model_batched.json, made by tests/tools/issue_model.py from the PMC summary it
names) and the line's own ms_per_step; the dose CRCs are the committed oracle CRCs (tests/golden/bench_dose_crc.json, made by
tests/golden/make_bench_crc.py with the ORACLE).  No GPU, no oracle call: arithmetic on committed files."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def line(name):
    with open(os.path.join(PROF, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_step_level_roofline_follows_from_the_pmc_model_and_the_timed_step():
    import bench
    d = line("r03/r03_final_bench_default.json")
    r = d["roofline"]
    with open(os.path.join(PROF, "extend_issue_model_batched.json")) as f:
        m = json.load(f)
    assert os.path.exists(os.path.join(ROOT, m["source"]))                    # the PMC summary the model was made from
    rays, seconds = d["config"]["rays_per_step"], d["ms_per_step"] * 1e-3
    util, bracket = bench.issue_model_utilisation(m, rays, seconds)
    for unit, v in util.items():
        assert r["step_level"][unit] == pytest.approx(v, abs=6e-4), unit
    assert r["frac"] == r["step_level"]["valu_issue"] == pytest.approx(r["achieved"] / r["peak"], abs=6e-4)
    assert r["frac_bracket"] == pytest.approx(bracket, abs=6e-4)
    assert r["level"].startswith("step")
    # per ray x rays / time, priced by hand: VALU issue cycles against 1024 SIMDs at 2.4 GHz
    by_hand = m["per_ray"]["valu_issue_cycles"] * rays / (1024 * 2.4e9 * seconds)
    assert r["frac"] == pytest.approx(by_hand, abs=6e-4)
    # the whole-job value is rays per step over the timed step
    assert d["value"] == pytest.approx(rays / seconds / 1e6, rel=2e-3)
    # the per-launch figures are the same model over the launch's own duration (two launches co-resident: not the headline)
    pl = r["per_launch"]
    u2, _ = bench.issue_model_utilisation(m, pl["rays_per_launch"], pl["avg_launch_ms"] * 1e-3)
    assert pl["utilisation_over_the_launch_wall_time"]["valu_issue"] == pytest.approx(u2["valu_issue"], abs=6e-4)
    # SURVEY 8d bookkeeping kept as the secondary figure
    h = r["hbm_algorithmic"]
    assert h["achieved_GBs_step_level"] == pytest.approx(h["algorithmic_bytes_per_ray"] * rays / seconds / 1e9, rel=2e-3)
    assert r["traffic"] == m["hbm_bytes"] and r["traffic_is"].startswith("static")


def test_kept_lines_carry_the_committed_oracle_crcs():
    with open(os.path.join(ROOT, "tests", "golden", "bench_dose_crc.json")) as f:
        crcs = json.load(f)
    flat = json.dumps(crcs)
    for name in ("r03/r03_final_bench_default.json", "r03/r03_final_bench_loop.json", "r03/r03_final_bench_loop_sync.json",
                 "r03/r03_final_bench_reference_semantics.json", "r03/r03_final_bench_route.json", "r03/r03_final_bench_route_loop_sync.json"):
        d = line(name)
        assert d["dose_crc32"] in flat, name
        assert d["higher_is_better"] is True and d["unit"] == "Mray/s" and d["vs_baseline"] is None
    d = line("r03/r03_final_bench_default.json")
    assert d["cpu_baseline"]["gpu_dose_bit_identical"] is True
    assert d["other_modes"]["reference_live_chain_semantics"]["dose_crc32"] in flat
    assert d["route_workload"]["dose_crc32"] in flat
    cold = d["cold_start"]
    assert cold["new_lamp_first_computation_ms"] <= 1.10 * cold["same_lamp_warm_ms"]


def test_bench_gpus_n_without_a_launcher_starts_children_and_reports_their_failure():
    """`python bench.py --gpus 2` with no WORLD_SIZE must not stop at "needs a launcher" (VERDICT r3): the parent starts
    two ranks; here (no GPU) both say so and exit non-zero, and the parent returns that status without hanging."""
    import subprocess
    import sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("the CPU-side failure path")
    except ImportError:
        pass
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "needs a GPU" in out.stderr and "torch.distributed.run" not in out.stderr
    assert "exited non-zero" in out.stderr or out.stderr.count("needs a GPU") == 2


def test_watchdog_ends_a_rank_that_waits_for_the_others_too_long():
    import subprocess
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog('rendezvous', 0.3):\n    time.sleep(30)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 3 and "stuck in 'rendezvous'" in out.stderr
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Watchdog('quick', 5):\n    pass\nprint('through')\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "through" in out.stdout
