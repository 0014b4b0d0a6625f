#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on one node: Mray/s of the UV-dose hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], mapped as SURVEY.md section 0 prescribes): rooms/C046_1.glb
is absent from the reference checkout, so its stand-in is the reference's only shipped scene,
testroomopt.glb (44 866 triangles vs 46 252); "1920x1080" = 2 073 600 photons per lamp launch;
"8-bounce" = 8 waves of the reference's one-segment photon pass (the reference has no bounces);
lamp 0 of positions/lange_route.xml; SEED_0 = 0.

One STEP = one whole computation by the reference's own host loop (RayTracer, myapp.cpp:156-175):
ResetDosageMap, then per wave generate -> extend -> accumulate and Shade (computeDosage +
dosageToColor), then -- with N > 1 -- the one reduction of the per-triangle maps over RCCL,
a final Shade and a sync.  Inputs (scene, BVH) are resident in HBM before the timed region.

The library pipelines consecutive launches over two HIP streams (include/uvrt.h uvrt_set_pipeline;
--no-pipeline turns it off), so inside the timed region the extend kernels of neighbouring waves
overlap and an event-bracketed kernel duration is not a kernel cost.  The `roofline` object is
therefore measured in a separate pass right after the timed region: the same step with the
pipelining off, HIP events around every uvrt_extend on its stream (`timing_pass`).

Scaling is WEAK: every GPU traces 8 waves; with N GPUs the computation has 8*N waves (launch k
runs on rank k % N, raytracer.h shardRank/shardWorld), so `value` = 8*N*2 073 600 rays / time.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PHOTONS = 1920 * 1080
WAVES = 8
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)


def census_path():
    return os.path.join(ROOT, "tests", "golden", "census_lamp0_%d.json" % PHOTONS)


def host_cores():
    """CPU cores this process may really use: affinity, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("UVRT_CPU_THREADS", "16"))))


def cpu_baseline(glb, route_xml, waves, photons, flavour=0):
    """The oracle (CPU restatement of the reference kernels, OpenMP on all host cores) timed on
    the same workload: `waves` launches of `photons` photons from lamp 0.  Also returns the
    traversal census that prices the algorithmic bytes per ray."""
    import __graft_entry__ as g
    orc = g.load_oracle()
    import numpy as np
    orc.set_flavour(flavour)
    s = orc.Scene(glb)
    r = orc.load_route(route_xml)
    cores = host_cores()
    c = orc.Computation(s, r["lamps"][:1], photons, r["lightHeight"], r["lightLength"], r["lightIntensity"],
                        nthreads=cores)
    c.reset()
    t_gen = t_ext = 0.0
    lamp = r["lamps"][0]
    lp = c.lamp_world_pos(lamp)
    t0 = time.time()
    for _ in range(waves):
        a = time.time()
        rays, c.SEED = orc.generate(0, photons, lp, c.lightLength, c.SEED)
        b = time.time()
        st = orc.extend(c.temp, s.tris, rays, s.nodes, s.triIdx, cores)
        d = time.time()
        orc.accumulate(c.photonMap, c.maxPhotonMap, c.temp, lamp[2])
        c.photonMapSize += photons
        c.stats.append(st)
        t_gen += b - a
        t_ext += d - b
    dose = c.dose()
    total = time.time() - t0
    tot = {k: sum(x[k] for x in c.stats) for k in ("rays", "aabb_tests", "tri_tests", "hits", "node_visits")}
    # The reference's own cl/extend.cl, compiled unmodified for gfx950 (oracle/_ref), timed on this
    # GPU on the last wave's rays: the closest thing to "the reference OpenCL path in the same run"
    # (no OpenCL CPU device exists in ROCm).
    ref_gpu = None
    try:
        if photons % 256 == 0 and orc.refgpu() is not None:
            rr = rays.copy()
            rr["dist"] = np.float32(1e30)
            rr["triID"] = 0
            _cnt, ms = orc.refgpu_extend(rr, s.tris, s.nodes, s.triIdx, reps=3)
            ref_gpu = {"kernel": "cl/extend.cl render, -O2, correctly rounded divide, no fast-math",
                       "ms_per_launch": round(ms, 3), "mray_s": round(photons / ms / 1e3, 1),
                       "triID_equal_to_oracle": float((rr["triID"] == rays["triID"]).mean())}
    except Exception as e:      # reporting only
        ref_gpu = {"error": str(e)[:200]}
    return {
        "reference_extend_cl_on_this_gpu": ref_gpu,
        "value": waves * photons / total / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
        "sample": "%d waves x %d photons, lamp 0 (the full step); generate is serial (defines the SEED "
                  "semantics), extend uses %d OpenMP threads; extend-only %.2f Mray/s"
                  % (waves, photons, cores, waves * photons / t_ext / 1e6),
    }, tot, dose


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--photons", type=int, default=PHOTONS)
    ap.add_argument("--waves", type=int, default=WAVES, help="waves (iterations) per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one stream: launches do not overlap")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): 8 waves per GPU, launches dealt to ranks, one SUM/MAX reduce per step; "
                         "strong: BASELINE configs[3] -- the same 8 waves split by global-id range over the ranks "
                         "('pixel tiles'), int32 count all-reduce per launch")
    ap.add_argument("--sort-bits", type=int, default=None)
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--flavour", type=int, default=0, choices=[0, 1],
                    help="arithmetic flavour of IntersectTri (include/uvrt.h uvrt_set_flavour): 0 = canonical strict "
                         "(SURVEY 8c), 1 = the fused cross/dot ROCm's OpenCL gives the reference's extend.cl on gfx950")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs one process per GPU: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    ndev = torch.cuda.device_count()
    rehearsal = world > ndev          # more ranks than GPUs: ranks share devices, collective over gloo
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import __graft_entry__ as g
    g.load_package()
    from uvrt_amd import host, sharding

    glb = os.path.join(ROOT, "tests", "golden", "testroomopt.glb")
    route_xml = os.path.join(ROOT, "tests", "golden", "lange_route.xml")

    # ---- product path: native loader + BVH + RayTracer on the HIP kernels -----------------
    rt = host.RayTracer(glb, route_xml, device=dev_index)
    rt.set_lamps(rt.lamps()[:1])              # lamp 0
    rt.photonCount = args.photons
    strong = args.scaling == "strong" and world > 1
    rt.maxIterations = args.waves if strong else args.waves * world
    rt.set_shard(0, 1) if strong else rt.set_shard(rank, world)
    # One real (non-default) torch stream carries BOTH the uvrt kernels and the collective, so the
    # reduction is ordered after the last accumulate and before the final Shade without host syncs.
    # (torch's default stream has handle 0, which uvrt_set_stream reads as "use your own stream".)
    stream = torch.cuda.Stream(device=device)
    assert stream.cuda_stream != 0
    rt.ctx.set_stream(stream.cuda_stream)
    torch.cuda.set_stream(stream)
    if args.sort_bits is not None:
        rt.ctx.set_sort_bits(args.sort_bits)
    if args.variant is not None:
        rt.ctx.set_variant(args.variant)
    if args.no_pipeline:
        rt.ctx.set_pipeline(False)
    rt.ctx.set_flavour(args.flavour)
    reducer = sharding.MapReducer(rt.ctx, device) if (world > 1 and not strong) else None

    if strong:
        # ray-range sharding of every launch: rank r generates and traces global ids
        # [r*n/world, (r+1)*n/world) (the global id feeds the seed, so the union is the unsharded
        # launch), the int32 per-triangle counts are summed over ranks, then every rank accumulates.
        n_launch = rt.photonsPerLight
        share = (n_launch + world - 1) // world
        first = min(rank * share, n_launch)
        mine = min(share, n_launch - first)
        lamp = rt.lamps()[0]
        lp = (lamp[0], float(np.float32(np.float32(rt.mesh.floorHeight) + np.float32(rt.lightHeight))), lamp[1])
        rt.ctx.set_pipeline(False)            # the count buffer is reduced in place after every launch:
        counts_t = sharding.wrap_array(rt.ctx, 2, device, "<i4")   # one buffer set, one stream

        def step():
            rt.ctx.seed = 0
            rt.ResetDosageMap()
            for _ in range(args.waves):
                rt.ctx.generate(lp, rt.lightLength, first, mine)
                rt.ctx.extend(mine)
                rt.ctx.device_ptr(2)                     # folds the deposit replicas into counts[0:T]
                dist.all_reduce(counts_t, op=dist.ReduceOp.SUM)
                rt.ctx.accumulate(lamp[2])
                rt.photonMapSize = rt.photonMapSize + n_launch
                rt.Shade()
                rt.currIterations = rt.currIterations + 1

    def _weak_step():
        rt.ctx.seed = 0                       # every step is the same computation (fresh-Init SEED)
        rt.ResetDosageMap()
        rt.set_shard(rank, world)             # restart the global launch index
        for it in range(rt.maxIterations):    # myapp.cpp:156-163
            rt.ComputeDosageMap()
            # one lamp => launch `it` belongs to rank it % world: every rank shades after each of ITS
            # launches (the per-GPU work of the N = 1 step), not after the launches it skipped
            if world == 1 or it % world == rank:
                rt.Shade()
            rt.currIterations = rt.currIterations + 1
        if reducer is not None:
            reducer()
            rt.Shade()

    if not strong:
        step = _weak_step

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    rt.Sync()                                  # surfaces a traversal-stack overflow, if any
    # the result of the TIMED region itself (the last of its steps), read before anything else runs
    import zlib
    dose_timed = rt.read_dosage()
    crc_timed = "%08x" % zlib.crc32(dose_timed.tobytes())
    # one computation on its own, bracketed by syncs (the reference syncs every iteration,
    # myapp.cpp:165; `value` is the steady-state rate of back-to-back computations)
    single_ms = []
    for _ in range(max(1, min(5, args.steps))):
        sync_all()
        t1 = time.perf_counter()
        step()
        sync_all()
        single_ms.append((time.perf_counter() - t1) * 1e3)
    single_ms.sort()
    single_ms = single_ms[len(single_ms) // 2]
    # timing pass for the roofline: the same step, launches not overlapped, events around extend
    timing_steps = max(1, min(3, args.steps))
    rt.ctx.set_pipeline(False)
    rt.ctx.set_timing(True)
    rt.ctx.extend_time_ms()                    # drop anything recorded so far
    for _ in range(timing_steps):
        step()
    sync_all()
    ext_ms, ext_launches = rt.ctx.extend_time_ms()
    rt.ctx.set_timing(False)
    rt.ctx.set_pipeline(not args.no_pipeline and not strong)
    rt.Sync()

    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    rays_per_step = rt.maxIterations * rt.photonsPerLight
    value = rays_per_step * args.steps / elapsed / 1e6
    dose = rt.read_dosage()
    ranks_agree = None
    if world > 1:
        # after the reduction every rank must hold the same maps, hence the same dose bits
        import zlib
        h = torch.tensor([zlib.crc32(dose.tobytes()), int(round(float(rt.ctx.read_photon_map(0).sum()) / 60.0))],
                         dtype=torch.int64, device=device)
        hs = [torch.zeros_like(h) for _ in range(world)]
        dist.all_gather(hs, h)
        ranks_agree = all(bool((x == hs[0]).all()) for x in hs)
        total_hits = int(hs[0][1].item())

    if rank == 0:
        # ---- CPU baseline + census (rank 0, N = 1 only) -----------------------------------
        cpu = None
        census = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, census, ref_dose = cpu_baseline(glb, route_xml, args.waves, args.photons, args.flavour)
            same = np.array_equal(dose_timed.view(np.uint32), ref_dose.view(np.uint32))
            cpu["gpu_dose_bit_identical"] = bool(same)
            cpu["checked"] = "dose read right after the timed (pipelined) steps, before the timing pass"
            if not same:
                raise SystemExit("bench: the dose of the timed region differs from the oracle's")
        if census is None and os.path.exists(census_path()) and args.photons == PHOTONS:
            census = json.load(open(census_path()))["per_launch_avg"]
            n_census = 1
        else:
            n_census = args.waves
        roof = None
        avg_ms = ext_ms / max(ext_launches, 1)
        model_path = os.path.join(ROOT, "profiles", "extend_issue_model.json")
        if os.path.exists(model_path):
            # The binding resource, priced with the per-ray instruction / lookup / byte counts of the kernel
            # (rocprofv3 PMC passes, deterministic per launch: profiles/extend_issue_model.json, made by
            # tests/tools/issue_model.py) and the issue rates calibrated on this GPU type
            # (tests/tools/valu_calib.hip, profiles/r02_valu_calibration.txt); the DURATION is this run's.
            m = json.load(open(model_path))
            k = m["constants"]
            pr = m["per_ray"]
            n_l = float(rt.photonsPerLight)
            sec = avg_ms * 1e-3
            scale = n_l / m["rays_per_launch"]
            simd_cycles = k["simds"] * k["clock_hz"] * sec
            cu_cycles = k["cus"] * k["clock_hz"] * sec
            util = {
                "valu_issue": pr["valu_issue_cycles"] * n_l / simd_cycles,
                "salu_issue": pr["salu_insts"] * n_l / cu_cycles,
                "l1_lookup": pr["l1_lane_lookups"] * n_l / (cu_cycles * k["l1_lookups_per_clk_per_cu"]),
                "hbm": pr["hbm_bytes"] * n_l / sec / k["hbm_peak_bytes_per_s"],
            }
            bound = max(util, key=util.get)
            vc = m["valu_issue_cycles"]
            roof = {"bound": bound, "kernel": "k_extend6<2,false,true,%s>" % ("true" if args.flavour else "false"),
                    "achieved": round(pr["valu_issue_cycles"] * n_l / sec / 1e9, 1),
                    "peak": round(k["simds"] * k["clock_hz"] / 1e9, 1), "unit": "G VALU issue-cycles/s",
                    "frac": round(util["valu_issue"], 4),
                    "frac_bracket": [round(vc["lower"] * scale / simd_cycles, 4), round(vc["upper"] * scale / simd_cycles, 4)],
                    "lane_utilisation": round(m["lane_utilisation"], 4),
                    "useful_lane_frac": round(util["valu_issue"] * m["lane_utilisation"], 4),
                    "utilisation_of_every_unit": {u: round(v, 4) for u, v in util.items()},
                    "wave_time_waiting_on_memory": round(m["wave_wait_frac"], 3) if m.get("wave_wait_frac") else None,
                    "traffic": round(pr["hbm_bytes"] * n_l),
                    "traffic_is": "static: PMC FETCH_SIZE x 2 + WRITE_SIZE of %s (not collected in this run)" % m["source"],
                    "rays_per_launch": rt.photonsPerLight, "avg_launch_ms": round(avg_ms, 4),
                    "extend_mray_s": round(rt.photonsPerLight / avg_ms / 1e3, 1),
                    "clock_assumed_ghz": k["clock_hz"] / 1e9,
                    "timing_pass": "%d step(s) with launch pipelining off after the timed region; HIP events "
                                   "around uvrt_extend on its stream" % timing_steps,
                    "model": "profiles/extend_issue_model.json (%s)" % m.get("note", ""),
                    "note": "no unit is saturated: the launch is latency-bound (waves spend about half their life in "
                            "s_waitcnt on record fetches) with VALU issue the busiest unit; packed f32 saves "
                            "instructions, not issue cycles (DESIGN.md 4)"}
        if census is not None:
            # secondary: SURVEY.md 8d's HBM-read figure (algorithmic bytes of the REFERENCE's layout, every node
            # visit priced as a memory read) -- exceeds the peak because the scene is L2/LDS resident
            n = float(census["rays"])
            bytes_per_ray = (32.0 + 8.0 + 32.0 * (1.0 + census["aabb_tests"] / n)
                             + 68.0 * census["tri_tests"] / n + 4.0 * census["hits"] / n)
            achieved = bytes_per_ray * rt.photonsPerLight / (avg_ms * 1e-3) / 1e9
            hbm_alg = {"algorithmic_bytes_per_ray": round(bytes_per_ray, 1), "achieved_GBs": round(achieved, 1),
                       "peak_GBs": HBM_PEAK_GBS, "frac": round(achieved / HBM_PEAK_GBS, 4),
                       "note": "SURVEY 8d bookkeeping only: > 1 because node visits are served by L2/LDS; measured HBM "
                               "traffic is `traffic` above"}
            if roof is None:
                roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None}
            roof["hbm_algorithmic"] = hbm_alg
        out = {
            "metric": "Mray/s (extend+shade) on C046_1.glb 1920x1080x8-bounce", "value": round(value, 2),
            "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "testroomopt.glb (stand-in for the absent rooms/C046_1.glb), %d photons/launch "
                                   "x %d waves per GPU, lamp 0 of lange_route.xml, SEED_0=0; step = reset + waves x "
                                   "(generate, extend, accumulate, shade)%s + sync"
                                   % (rt.photonsPerLight, args.waves, " + RCCL SUM/MAX of the per-triangle maps"
                                      if world > 1 else ""),
                       "triangles": rt.mesh.triangleCount, "rays_per_step": rays_per_step,
                       "launch_pipelining": bool(not args.no_pipeline and not strong),
                       "parallelism": "launch-sharded x%d" % world + (" (REHEARSAL: ranks share a GPU, gloo)" if rehearsal else "")},
            "roofline": roof, "cpu_baseline": cpu,
        }
        out["dose_crc32"] = crc_timed
        out["dose_crc32_after_timing_pass"] = "%08x" % zlib.crc32(dose.tobytes())
        if out["dose_crc32"] != out["dose_crc32_after_timing_pass"]:
            raise SystemExit("bench: pipelined and one-stream passes disagree (%s vs %s)"
                             % (out["dose_crc32"], out["dose_crc32_after_timing_pass"]))
        out["value_is"] = "steady-state throughput of back-to-back computations (no sync between steps)"
        out["single_computation"] = {"ms": round(single_ms, 4), "mray_s": round(rays_per_step / single_ms / 1e3, 1),
                                     "note": "one step bracketed by device syncs, median of %d" % max(1, min(5, args.steps))}
        out["config"]["flavour"] = args.flavour
        if world > 1:
            out["multi_gpu_check"] = {"dose_identical_on_all_ranks": ranks_agree, "photons_deposited": total_hits,
                                      "photons_traced": rays_per_step}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    rt.close()


if __name__ == "__main__":
    main()
