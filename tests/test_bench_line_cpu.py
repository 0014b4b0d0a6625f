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


def test_step_level_roofline_follows_from_the_census_model_the_measured_clock_and_the_timed_step():
    """Every fraction of the kept line's `roofline` re-derived from committed files: the per-ray model
    (profiles/extend_issue_model_batched.json = tests/tools/stream_census.py over the trip census and the PMC summary it names),
    the line's own ms_per_step and the shader clock the run measured."""
    import bench
    d = line("r04/r04_final_bench_default.json")
    r = d["roofline"]
    with open(os.path.join(PROF, "extend_issue_model_batched.json")) as f:
        m = json.load(f)
    assert os.path.exists(os.path.join(ROOT, m["source"])) and os.path.exists(os.path.join(ROOT, m["census"]))
    rays, seconds = d["config"]["rays_per_step"], d["ms_per_step"] * 1e-3
    clock = r["clock_measured_mhz"] * 1e6
    assert 1.5e9 < clock < 2.6e9                                              # measured in the run, not the nominal constant
    util, bracket = bench.issue_model_utilisation(m, rays, seconds, clock)
    for unit, v in util.items():
        assert r["step_level"][unit] == pytest.approx(v, abs=1e-3), unit
    assert set(r["step_level"]) == {"valu_issue", "salu_issue", "l1_lookup", "deposit_atomics", "hbm"}
    assert r["frac"] == r["step_level"]["valu_issue"] == pytest.approx(r["achieved"] / r["peak"], abs=2e-3)
    assert r["frac_bracket"] == pytest.approx(bracket, abs=1e-3)
    assert r["bound"] == max(util, key=util.get) and r["level"].startswith("step")
    # the bracket is the +-15 % on the compiler-written quarter of the instructions: within +-5 % of the estimate
    assert 0.95 < r["frac_bracket"][0] / r["frac"] < 1.0 < r["frac_bracket"][1] / r["frac"] < 1.05
    # by hand: VALU issue cycles per ray x rays over 1024 SIMDs at the measured clock
    by_hand = m["per_ray"]["valu_issue_cycles"] * rays / (1024 * clock * seconds)
    assert r["frac"] == pytest.approx(by_hand, abs=1e-3)
    assert r["frac_at_nominal_2400mhz"] == pytest.approx(m["per_ray"]["valu_issue_cycles"] * rays / (1024 * 2.4e9 * seconds), abs=1e-3)
    # deposits: scattered atomic adds against the calibrated rate (profiles/r04/r04_atomic_calibration.txt)
    rate = float(open(os.path.join(PROF, "r04", "r04_atomic_calibration.txt")).read().split("8 workgroups per CU:")[1].split("=")[1].split("G")[0]) * 1e9
    assert m["constants"]["scattered_atomic_adds_per_s"] == pytest.approx(rate, rel=0.02)
    assert r["step_level"]["deposit_atomics"] == pytest.approx(m["per_ray"]["deposit_atomics"] * rays / seconds / m["constants"]["scattered_atomic_adds_per_s"], abs=1e-3)
    # the model itself: stream trips x static cycles per kind + the rest, as the census tool writes it
    st = m["stream_per_ray"]["valu_cycles"]
    rest = m["compiler_written"]["valu_insts_per_ray"] * m["compiler_written"]["static_mean_cycles"]
    assert m["per_ray"]["valu_issue_cycles"] == pytest.approx(st + rest, rel=1e-9)
    kinds = m["per_trip_kind"]
    assert st == pytest.approx(sum(m["trips_per_ray"][k] * kinds[k]["valu_cycles"] for k in m["trips_per_ray"]), rel=1e-9)
    assert kinds["stream_in"]["valu"] == 56 and kinds["stream_in"]["valu_cycles"] == 218      # the inner-node trip of the strict stream
    # the whole-job value is rays per step over the timed step
    assert d["value"] == pytest.approx(rays / seconds / 1e6, rel=2e-3)
    pl = r["per_launch"]
    u2, _ = bench.issue_model_utilisation(m, pl["rays_per_launch"], pl["avg_launch_ms"] * 1e-3, clock)
    assert pl["utilisation_over_the_launch_wall_time"]["valu_issue"] == pytest.approx(u2["valu_issue"], abs=1e-3)
    h = r["hbm_algorithmic"]
    assert h["achieved_GBs_step_level"] == pytest.approx(h["algorithmic_bytes_per_ray"] * rays / seconds / 1e9, rel=2e-3)
    assert r["traffic_is"].startswith("static")


def test_static_tally_of_the_hand_written_stream():
    """tests/tools/stream_census.py expands the R7_* macros of uvrt_extend6.hip through the C preprocessor and prices every
    instruction with the calibrated classes: the inner-node trip is 56 VALU instructions = 218 issue cycles in the exact
    flavours and 44 = 170 with the shipped-flags slab arithmetic (the 12 packed fmas of the exact division gone)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import stream_census as sc
    for fl, (valu, cyc) in {0: (56, 218), 1: (56, 218), 2: (44, 170)}.items():
        seg = {k: sc.tally(v) for k, v in sc.segments(sc.stream_text(fl)).items()}
        inner = [sum(seg[s][f] for s in "ACEF") for f in ("valu", "valu_cycles")]
        assert inner == [valu, cyc], (fl, inner)
        assert seg["C"]["vmem"] == 4 and seg["C"]["lds"] == 5               # four record loads from memory, four from LDS + the stack top
        assert seg["D"]["by_cost"][8] == 1                                  # one v_rcp_f32 per triangle block


def test_kept_lines_carry_the_committed_oracle_crcs():
    with open(os.path.join(ROOT, "tests", "golden", "bench_dose_crc.json")) as f:
        crcs = json.load(f)
    flat = json.dumps(crcs)
    for name in ("r04/r04_final_bench_default.json", "r04/r04_final_bench_loop_sync.json", "r03/r03_final_bench_loop.json",
                 "r03/r03_final_bench_reference_semantics.json", "r03/r03_final_bench_route.json", "r03/r03_final_bench_route_loop_sync.json"):
        d = line(name)
        assert d["dose_crc32"] in flat, name
        assert d["higher_is_better"] is True and d["unit"] == "Mray/s" and d["vs_baseline"] is None
    d = line("r04/r04_final_bench_default.json")
    assert d["cpu_baseline"]["gpu_dose_bit_identical"] is True
    # the opt-in shipped-flags flavour: its dose CRC equals the oracle's in that flavour, computed in the same run
    f2 = d["other_modes"]["shipped_flags_flavour"]
    assert f2["dose_crc32"] == f2["dose_crc32_expected"] and f2["dose_crc32"] != d["dose_crc32"]
    assert f2["triangles_beyond_1e-4_of_flavour0"] < 50
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
