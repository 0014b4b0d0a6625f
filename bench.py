#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on one node: Mray/s of the UV-dose hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2] at N = 1, configs[3] at N > 1, mapped as SURVEY.md section 0 prescribes):
rooms/C046_1.glb is absent from the reference checkout, so its stand-in is the reference's only shipped scene,
testroomopt.glb (44 866 triangles vs 46 252); "1920x1080" = 2 073 600 photons per lamp launch; "8-bounce" =
8 waves (iterations) of the reference's one-segment photon pass (the reference has no bounces); lamp 0 of
positions/lange_route.xml; SEED_0 = 0.  `--scene soup:T` swaps in a synthetic T-triangle scene that does not
fit in L2.  `--route` makes the reference's own DEFAULT workload the headline instead: all 12 lamps
of lange_route.xml x 10 iterations x ((2^25 / 12) & ~1) photons (raytracer.h:30-32, myapp.cpp:156-170); the
default run reports it beside the headline (`route_workload`).

One STEP = one whole computation: ResetDosageMap, then per wave generate -> extend -> accumulate and Shade
(computeDosage + dosageToColor), then a sync.  Inputs (scene, BVH) are resident in HBM before the timed region.
Modes at N = 1 (`value` is the selected one's, the others are reported under `other_modes`):
  batched (default)   RayTracer::ComputeIterationsBatched -- an API EXTENSION the reference's caller does not have
                      (include/uvrt.h "batched tracing"): the waves traced first in fused launches of a few waves
                      each on two side streams, then accumulate + Shade replayed per wave on the context's stream
                      while the next computation is already being traced: the same arithmetic, the same bits;
  loop                the reference's host loop call by call (myapp.cpp:156-163) with the library's two-stream
                      launch pipelining, no host sync inside a computation;
  loop_sync           the same loop with the reference's clFinish after every iteration (myapp.cpp:165): what an
                      unmodified MyApp::Tick sees.

N > 1 (one process per GPU): STRONG scaling is the headline -- BASELINE configs[3]: every launch is split by
global-id range over the ranks ("pixel tiles"), each rank deposits into private int32 planes, ONE RCCL
all-reduce of the planes per computation (native: uvrt_reduce_batch), then every rank replays accumulate +
Shade; the dose is bit-identical to one GPU.  The WEAK figure (configs[4]: whole launches dealt to ranks, 8
waves per GPU, one SUM/MAX reduction of the f64 maps) is measured in the same run and reported beside it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PHOTONS = 1920 * 1080
WAVES = 8
ROUTE_PHOTONS = 1 << 25      # raytracer.h:30 photonCount
ROUTE_ITERATIONS = 10        # raytracer.h:32 maxIterations
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


def cpu_baseline(scene, route, waves, photons, flavour=0, budget_s=25.0):
    """The oracle (CPU restatement of the reference kernels, OpenMP on the host's cores) timed on the same
    workload: `waves` launches of `photons` photons from lamp 0, cut short after about `budget_s` seconds of CPU
    work.  Also returns the traversal census that prices the algorithmic bytes per ray and, when
    all waves were run, the reference dose."""
    import __graft_entry__ as g
    orc = g.load_oracle()
    import numpy as np
    orc.set_flavour(flavour)
    s, r = scene, route
    cores = host_cores()
    c = orc.Computation(s, r["lamps"][:1], photons, r["lightHeight"], r["lightLength"], r["lightIntensity"],
                        nthreads=cores)
    c.reset()
    t_gen = t_ext = 0.0
    lamp = r["lamps"][0]
    lp = c.lamp_world_pos(lamp)
    t0 = time.time()
    done = 0
    for w in range(waves):
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
        done += 1
        if time.time() - t0 > budget_s and w + 1 < waves:
            break
    total = time.time() - t0
    dose = c.dose() if done == waves else None
    tot = {k: sum(x[k] for x in c.stats) for k in ("rays", "aabb_tests", "tri_tests", "hits", "node_visits")}
    # The reference's own cl/extend.cl, compiled unmodified for gfx950 (oracle/_ref), timed on this GPU on the
    # last wave's rays: the closest thing to "the reference OpenCL path in the same run" (no OpenCL CPU device
    # exists in ROCm).
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
            if orc.refgpu_have_shipped():
                # the same source with the reference's OWN clBuildProgram options (template/template.cpp:1192):
                # -cl-fast-relaxed-math -cl-mad-enable -cl-single-precision-constant -- the fairer same-GPU baseline
                rs = rays.copy()
                rs["dist"] = np.float32(1e30)
                rs["triID"] = 0
                _cnt2, ms2 = orc.refgpu_extend(rs, s.tris, s.nodes, s.triIdx, reps=3, shipped=True)
                ref_gpu["as_shipped"] = {
                    "kernel": "cl/extend.cl render with the reference's own flags: -cl-fast-relaxed-math -cl-mad-enable "
                              "-cl-single-precision-constant (template.cpp:1192)",
                    "ms_per_launch": round(ms2, 3), "mray_s": round(photons / ms2 / 1e3, 1),
                    "triID_equal_to_the_strict_build": float((rs["triID"] == rr["triID"]).mean()),
                    "rays_on_another_triangle_than_the_strict_build": int((rs["triID"] != rr["triID"]).sum())}
    except Exception as e:      # reporting only
        ref_gpu = {"error": str(e)[:200]}
    orc.set_flavour(0)
    ocl = opencl_cpu_devices()
    return {
        "opencl_host_cpu_device": ("no OpenCL loader" if ocl is None else
                                   "%d platform(s), %d CPU device(s)%s" % (ocl[0], ocl[1], "" if ocl[1] else
                                                                            ": the reference's OpenCL path cannot run on the host CPU here; kind = port")),
        "reference_extend_cl_on_this_gpu": ref_gpu,
        "value": done * photons / total / 1e6, "unit": "Mray/s", "cores": cores, "kind": "port",
        "sample": "%d of the step's %d waves x %d photons, lamp 0; generate is serial (defines the SEED "
                  "semantics), extend uses %d OpenMP threads; extend-only %.2f Mray/s"
                  % (done, waves, photons, cores, done * photons / t_ext / 1e6),
    }, tot, dose


def oracle_dose(scene, route, waves, photons, flavour):
    """Checker for a leg whose arithmetic has no committed CRC: the oracle's dose of the whole step (lamp 0, `waves`
    launches of `photons`) in `flavour`.  Flavour 2 evaluates v_rcp_f32 through the table read from THIS GPU
    (oracle/rcp_model.h), which is why its CRC cannot be a committed fixture."""
    import __graft_entry__ as g
    orc = g.load_oracle()
    orc.set_flavour(flavour)
    try:
        c = orc.Computation(scene, route["lamps"][:1], photons, route["lightHeight"], route["lightLength"],
                            route["lightIntensity"], nthreads=host_cores())
        c.reset()
        for _ in range(waves):
            c.iteration()
        return c.dose()
    finally:
        orc.set_flavour(0)


_OCL_PROBE = r"""
import ctypes
cl = None
for name in ("libOpenCL.so.1", "libOpenCL.so", "/opt/rocm/lib/libOpenCL.so.1"):
    try:
        cl = ctypes.CDLL(name)
        break
    except OSError:
        pass
if cl is None:
    print("none")
else:
    n = ctypes.c_uint(0)
    cpus = 0
    if cl.clGetPlatformIDs(0, None, ctypes.byref(n)) == 0 and n.value:
        plats = (ctypes.c_void_p * n.value)()
        cl.clGetPlatformIDs(n.value, plats, None)
        for pl in plats:
            m = ctypes.c_uint(0)
            if cl.clGetDeviceIDs(ctypes.c_void_p(pl), ctypes.c_uint64(2), 0, None, ctypes.byref(m)) == 0:   # CL_DEVICE_TYPE_CPU
                cpus += m.value
    print(n.value, cpus)
"""


def opencl_cpu_devices():
    """the north_star asks for the reference's OpenCL path on a host-CPU OpenCL device.  Enumerate what the
    OpenCL ICD loader offers: (platforms, CPU devices); None when no loader can be opened.  Done in a child process (a
    second runtime in this one is not worth a risk to the line).  ROCm's runtime has no CPU device, here or on the GPU
    box; nothing is run either way -- the figure documents why the CPU baseline is the port."""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", _OCL_PROBE], capture_output=True, text=True, timeout=30).stdout.split()
        return (int(out[0]), int(out[1])) if len(out) == 2 else None
    except Exception:
        return None


def soup_triangles(T, seed=1):
    """A synthetic scene that does not fit in L2: T small random triangles filling the test room's bounding box."""
    import numpy as np
    rng = np.random.default_rng(seed)
    lo = np.array([-1.68, -1.44, -3.73], dtype=np.float32)
    hi = np.array([1.63, 1.31, 5.56], dtype=np.float32)
    c = rng.uniform(lo, hi, (T, 3)).astype(np.float32)
    size = np.float32(0.5 * (float(np.prod(hi - lo)) / T) ** (1.0 / 3.0))
    tris = np.zeros((T, 16), dtype=np.float32)
    for k in range(3):
        tris[:, 4 * k:4 * k + 3] = c + rng.uniform(-size, size, (T, 3)).astype(np.float32)
    return tris


class _Scene:
    pass


def issue_model_utilisation(m, rays, seconds, clock_hz=None):
    """Utilisation of each unit while `rays` rays are traced in `seconds` of wall time: the kernel's per-ray counts
    (profiles/extend_issue_model_*.json, made by tests/tools/stream_census.py: VALU issue cycles = trip counts x the static
    cycles of the hand-written stream per kind of trip + the compiler-written rest from the PMC instruction count; scalar
    instructions, L1 lookups, fabric bytes and deposit atomics from rocprofv3 PMC passes -- all deterministic per launch)
    over the unit peaks: issue slots at `clock_hz`, the shader clock MEASURED during the run (uvrt_clock_probe_*; nominal if
    None), L1 lookups and scattered atomics at their calibrated rates (tests/tools/fetch_calib.hip, atomic_calib.hip).
    Returns (utilisation per unit, [lower, upper] of the VALU figure)."""
    k, pr = m["constants"], m["per_ray"]
    clock = clock_hz or k["clock_hz_nominal"]
    simd_cycles = k["simds"] * clock * seconds
    cu_cycles = k["cus"] * clock * seconds
    util = {
        "valu_issue": pr["valu_issue_cycles"] * rays / simd_cycles,
        "salu_issue": pr["salu_insts"] * rays / cu_cycles,
        "l1_lookup": pr["l1_lane_lookups"] * rays / (cu_cycles * k["l1_lookups_per_clk_per_cu"]),
        "deposit_atomics": pr["deposit_atomics"] * rays / seconds / k["scattered_atomic_adds_per_s"],
        "hbm": pr["hbm_bytes"] * rays / seconds / k["hbm_peak_bytes_per_s"],
    }
    lo, hi = pr["valu_issue_cycles_bracket"]
    return util, [lo * rays / simd_cycles, hi * rays / simd_cycles]


class Watchdog:
    """A rank that does not get through `stage` (a rendezvous, a collective init: calls that wait for OTHER ranks) within
    `seconds` says so and exits non-zero; the parent (spawn_ranks / torch.distributed.run) then stops the others.  The
    process ends, nothing is re-exec'd."""

    def __init__(self, stage, seconds=None):
        import threading
        self.stage = stage
        self.seconds = float(os.environ.get("UVRT_RENDEZVOUS_TIMEOUT_S", "300")) if seconds is None else seconds
        self.timer = threading.Timer(self.seconds, self.fire)
        self.timer.daemon = True

    def fire(self):
        sys.stderr.write("bench.py: rank %s stuck in '%s' for %.0f s -- exiting 3\n"
                         % (os.environ.get("RANK", "0"), self.stage, self.seconds))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def spawn_ranks(n):
    """Start `n` ranks of this script as CHILD processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, the same command line), relay rank 0's stdout, and return the exit status: 0 only if every rank
    exited 0.  The caller has not touched the GPU -- nothing is re-exec'd.  A rank that dies takes the others with it
    (they would wait for it in a collective for ever); a job that outlives UVRT_BENCH_TIMEOUT_S (default 1500) is
    killed rank by rank -- the exact PIDs started here -- and the status is non-zero, with the ranks still running named."""
    import socket
    import subprocess
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(sk.getsockname()[1])
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    limit = float(os.environ.get("UVRT_BENCH_TIMEOUT_S", "1500"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.time()
    status = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                status = next((c for c in codes if c), 0)
                break
            failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
            late = time.time() - t0 > limit
            if failed or late:
                running = [r for r, c in enumerate(codes) if c is None]
                sys.stderr.write("bench.py --gpus %d: %s; stopping rank(s) %s\n"
                                 % (n, ("rank(s) %s exited non-zero" % failed) if failed else
                                    ("no result after %.0f s" % limit), running))
                status = next((c for c in codes if c), 1) if failed else 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--photons", type=int, default=PHOTONS)
    ap.add_argument("--waves", type=int, default=WAVES, help="waves (iterations) of the computation")
    ap.add_argument("--mode", choices=["batched", "loop", "loop_sync"], default="batched",
                    help="N = 1: batched (default) = RayTracer::ComputeIterationsBatched (an API extension); loop = the "
                         "reference's host loop call by call; loop_sync = that loop with the reference's device sync after "
                         "every iteration (myapp.cpp:165).  N > 1 always runs the batched, sharded computation")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1 headline: strong (default, BASELINE configs[3]: launches split by global-id range, one "
                         "int32 all-reduce of the count planes per computation) or weak (configs[4]: whole launches dealt "
                         "to ranks, 8 waves per GPU); the other one is measured too and reported beside it")
    ap.add_argument("--scene", default=None, help="a .glb file, or soup:T for a synthetic T-triangle scene")
    ap.add_argument("--route", action="store_true",
                    help="headline = the reference's default workload: 12 lamps x 10 iterations x ((2^25 / 12) & ~1) photons "
                         "(raytracer.h:30-32, positions/lange_route.xml, myapp.cpp:156-170)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lean", action="store_true",
                    help="profiling aid: only the headline mode (no other-mode, single-computation, cold or route leg), so "
                         "that a kernel trace / PMC run holds nothing but the timed kind of launch")
    ap.add_argument("--no-pipeline", action="store_true", help="one stream: launches do not overlap")
    ap.add_argument("--sort-bits", type=int, default=None)
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--wide", action="store_true",
                    help="opt-in 4-wide BVH walk (include/uvrt.h uvrt_set_wide_bvh): not the reference's visit order; the "
                         "dose check against the oracle still applies (it fails on a scene with order-dependent rays)")
    ap.add_argument("--flavour", type=int, default=0, choices=[0, 1, 2],
                    help="arithmetic flavour of extend (include/uvrt.h uvrt_set_flavour): 0 = canonical strict "
                         "(SURVEY 8c), 1 = the fused cross/dot ROCm's OpenCL gives the reference's extend.cl on gfx950, "
                         "2 = OPT-IN \"shipped flags\": the arithmetic of extend.cl built with the reference's own "
                         "-cl-fast-relaxed-math flags on gfx950 (slab test by v_rcp_f32 and a multiply)")
    ap.add_argument("--seed-mode", type=int, default=0, choices=[0, 1],
                    help="SEED semantics of generate.cl (include/uvrt.h uvrt_set_seed_mode): 0 = canonical (SURVEY 8c), "
                         "1 = what the reference's kernel does on gfx950; with --flavour 1 the reference's live kernel chain")
    ap.add_argument("--self-comm", action="store_true",
                    help="N = 1 rehearsal of a rank's step of the sharded job: a one-rank RCCL communicator, so that every "
                         "computation launches the real all-reduce kernel of the count planes (uvrt_reduce_batch) between its "
                         "tracing and its replay; with --photons 259200 the step is a rank's share of the 8-GPU step")
    ap.add_argument("--comm", choices=["native", "torch"], default="native",
                    help="N > 1 (and --self-comm): who all-reduces the count planes -- native = the context's own RCCL "
                         "communicator (uvrt_reduce_batch), torch = torch.distributed's RCCL all-reduce of the same device "
                         "planes (the fallback the run takes by itself when the native communicator is unavailable on any rank)")
    ap.add_argument("--high-priority-stream", action="store_true",
                    help="the context's stream (replay, collective) is created with high priority; the launch lanes keep the default")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: this process has made no GPU call (torch is not even imported yet); it
        # starts one fresh child per rank, relays rank 0's line and exits with the children's status
        raise SystemExit(spawn_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    if world > 1 and args.mode != "batched":
        raise SystemExit("--mode %s is a single-GPU mode (N > 1 runs the sharded batched computation)" % args.mode)
    if args.route and (args.scene or world > 1):
        raise SystemExit("--route is the single-GPU test-room workload")
    ndev = torch.cuda.device_count()
    rehearsal = world > ndev          # more ranks than GPUs: ranks share devices, collectives over gloo
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    self_torch = args.self_comm and args.comm == "torch"     # N = 1 rehearsal of the torch-fallback branch
    if world > 1 or self_torch:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if self_torch:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        with Watchdog("torch.distributed rendezvous (%s)" % ("gloo" if rehearsal else "nccl")):
            if rehearsal:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    coll_device = "cpu" if rehearsal else device

    import __graft_entry__ as g
    g.load_package()
    from uvrt_amd import capi, host, sharding

    glb = os.path.join(ROOT, "tests", "golden", "testroomopt.glb")
    route_xml = os.path.join(ROOT, "tests", "golden", "lange_route.xml")
    scene_label = "testroomopt.glb (stand-in for the absent rooms/C046_1.glb)"
    oscene = None
    if args.scene and args.scene.startswith("soup:"):
        T_soup = int(args.scene[5:])
        mesh = host.Mesh(tris=soup_triangles(T_soup))     # BVH by the native builder (host/bvh.cpp)
        rt = host.RayTracer(None, route_xml, device=dev_index, mesh=mesh)
        scene_label = "synthetic soup of %d triangles in the test room's box (beyond L2)" % T_soup
        oscene = _Scene()
        oscene.tris, oscene.nodes, oscene.triIdx, oscene.T = mesh.tris(), mesh.nodes(), mesh.triIdx(), T_soup
        oscene.floorHeight = np.float32(mesh.floorHeight)
    else:
        if args.scene:
            glb = args.scene
            scene_label = os.path.basename(glb)
        rt = host.RayTracer(glb, route_xml, device=dev_index)
    all_lamps = rt.lamps()
    if args.route:
        args.photons, args.waves = ROUTE_PHOTONS, ROUTE_ITERATIONS
    default_config = (args.scene is None and args.photons == PHOTONS and args.waves == WAVES and not args.route)

    def configure(lamps, photon_count, iterations):
        rt.set_lamps(lamps)
        rt.photonCount = photon_count
        rt.maxIterations = iterations

    headline_lamps = all_lamps if args.route else all_lamps[:1]
    configure(headline_lamps, args.photons, args.waves)
    # One real (non-default) torch stream carries BOTH the uvrt kernels and any torch collective, so reductions
    # are ordered after the last deposit and before the replay without host syncs.
    stream = torch.cuda.Stream(device=device, priority=-1 if args.high_priority_stream else 0)
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
    rt.ctx.set_seed_mode(args.seed_mode)
    if args.wide:
        rt.ctx.set_wide_bvh(True)
    n_launch = rt.photonsPerLight
    n_lamps = len(headline_lamps)
    lamp = headline_lamps[0]
    lp = (lamp[0], float(np.float32(np.float32(rt.mesh.floorHeight) + np.float32(rt.lightHeight))), lamp[1])

    # ---- the sharded (strong) computation: ray ranges + one collective per batch ---------------------------
    native_comm = False
    comm_note = ""
    comm_info = None
    if world > 1 and not rehearsal and args.comm == "native":
        # the context's own RCCL communicator (uvrt_comm_init_rank); should it be unavailable on ANY rank, every rank
        # reduces the planes with torch.distributed's RCCL all-reduce on the device instead (same collective, one more
        # host call).  ncclCommInitRank is itself a collective, so the ranks first agree on a LOCAL precondition
        # (librccl opens, its symbols resolve, rank 0 got an id): nobody enters the init unless everybody will.
        ok, why = capi.comm_available()
        ids = [None]
        if ok and rank == 0:
            try:
                ids = [capi.comm_unique_id()]
            except Exception as e:          # noqa: BLE001  (reported in the line)
                ok, why = False, str(e)
        flag = torch.tensor([int(ok)], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            dist.broadcast_object_list(ids, src=0)
            ok = True
            try:
                with Watchdog("uvrt_comm_init_rank (ncclCommInitRank, %d ranks)" % world):
                    rt.ctx.comm_init_rank(ids[0], rank, world)
            except Exception as e:      # noqa: BLE001
                ok, why = False, str(e)
            flag = torch.tensor([int(ok)], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native_comm = bool(int(flag.item()))
            if not native_comm and ok:
                rt.ctx.comm_destroy()
        if not native_comm:
            comm_note = "native communicator unavailable (%s): torch.distributed RCCL all-reduce of the planes" % (why or "another rank failed")
    elif world > 1 and not rehearsal:
        comm_note = "--comm torch: torch.distributed RCCL all-reduce of the planes"
    if args.self_comm:
        if world != 1 or args.mode != "batched":
            raise SystemExit("--self-comm is the N = 1 rehearsal of the batched, sharded step")
        if not self_torch:
            rt.ctx.comm_init_rank(capi.comm_unique_id(), 0, 1)
            rt.set_reduce_over_comm(True)
    if native_comm or (args.self_comm and not self_torch):
        comm_info = rt.ctx.comm_info()
        if comm_info["rccl_ranks"] != world:
            raise SystemExit("bench: RCCL says the communicator spans %d ranks, the job has %d" % (comm_info["rccl_ranks"], world))
    torch_reduce = (world > 1 and not rehearsal and not native_comm) or self_torch
    share = (n_launch + world - 1) // world
    first = min(rank * share, n_launch)
    mine = min(share, n_launch - first)

    def step_batched():
        rt.ctx.seed = 0                       # every step is the same computation (fresh-Init SEED)
        rt.ResetDosageMap()
        if torch_reduce:
            # the same step with torch.distributed's RCCL all-reduce of the device planes (the stream is torch's current one)
            ops = np.zeros(args.waves, dtype=capi.REPLAY_OP_DT)
            for k in range(args.waves):
                ops[k] = (lamp[2], 1, 0, (k + 1) * n_launch, float(np.float32(rt.lightIntensity) * np.float32(0.1)),
                          rt.minDosage, 0)
            rt.ctx.trace_batch([lp] * args.waves, rt.lightLength, first, mine)
            dist.all_reduce(sharding.wrap_array(rt.ctx, 5, device, "<i4"), op=dist.ReduceOp.SUM)
            rt.ctx.replay_batch(ops)
            rt.photonMapSize = args.waves * n_launch
            rt.currIterations = args.waves
        elif world > 1 and rehearsal:
            # ranks share a GPU: RCCL wants one rank per device, so the planes go through gloo (test path only)
            ops = np.zeros(args.waves, dtype=capi.REPLAY_OP_DT)
            for k in range(args.waves):
                ops[k] = (lamp[2], 1, 0, (k + 1) * n_launch, float(np.float32(rt.lightIntensity) * np.float32(0.1)),
                          rt.minDosage, 0)
            rt.ctx.trace_batch([lp] * args.waves, rt.lightLength, first, mine)
            planes = sharding.wrap_array(rt.ctx, 5, device, "<i4")
            torch.cuda.current_stream().synchronize()
            host_planes = planes.cpu()
            dist.all_reduce(host_planes, op=dist.ReduceOp.SUM)
            planes.copy_(host_planes)
            rt.ctx.replay_batch(ops)
            rt.photonMapSize = args.waves * n_launch
            rt.currIterations = args.waves
        else:
            rt.ComputeIterationsBatched(rt.maxIterations)

    def step_loop():
        rt.ctx.seed = 0
        rt.ResetDosageMap()
        for _ in range(rt.maxIterations):     # myapp.cpp:156-163
            rt.ComputeDosageMap()
            rt.Shade()
            rt.currIterations = rt.currIterations + 1

    def step_loop_sync():
        rt.ctx.seed = 0
        rt.ResetDosageMap()
        for _ in range(rt.maxIterations):     # myapp.cpp:156-165: clFinish after every iteration
            rt.ComputeDosageMap()
            rt.Shade()
            rt.currIterations = rt.currIterations + 1
            rt.Sync()

    STEP = {"batched": step_batched, "loop": step_loop, "loop_sync": step_loop_sync}
    reducer = [None]

    def step_weak():
        # whole launches dealt to ranks: 8 waves per GPU, launch k on rank k % world, one SUM/MAX of the maps
        rt.ctx.seed = 0
        rt.ResetDosageMap()
        rt.set_shard(rank, world)
        for it in range(args.waves * world):
            rt.ComputeDosageMap()
            if it % world == rank:
                rt.Shade()
            rt.currIterations = rt.currIterations + 1
        reducer[0]()
        rt.Shade()

    def sync_all():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    def timed(step, warmup, steps):
        for _ in range(warmup):
            step()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync_all()
        el = time.perf_counter() - t0
        rt.Sync()                              # surfaces a traversal-stack overflow, if any
        if world > 1:
            tmax = torch.tensor([el], dtype=torch.float64, device=coll_device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        return el

    def one_synced(step):
        sync_all()
        t1 = time.perf_counter()
        step()
        sync_all()
        return (time.perf_counter() - t1) * 1e3

    def median_synced(step, k):
        sm = sorted(one_synced(step) for _ in range(k))
        return sm[len(sm) // 2]

    def crc(a):
        return "%08x" % zlib.crc32(a.tobytes())

    golden_path = os.path.join(ROOT, "tests", "golden", "bench_dose_crc.json")
    golden = json.load(open(golden_path)) if os.path.exists(golden_path) else {}

    rays_per_step = args.waves * n_launch * n_lamps
    other_modes = {}
    weak = strong = None
    single_ms = None
    ext_ms = ext_launches = None
    clock_mhz = None
    cold = route_leg = None
    timing_steps = max(1, min(3, args.steps))
    few = max(1, min(5, args.steps))

    if world == 1:
        headline = STEP[args.mode]
        elapsed = timed(headline, args.warmup, args.steps)
        # the result of the TIMED region itself (the last of its steps), read before anything else runs
        dose_timed = rt.read_dosage()
        if not args.lean:
            # one computation on its own, bracketed by syncs (`value` is the steady-state rate of back-to-back
            # computations)
            single_ms = median_synced(headline, few)
            # the other modes, for comparison, and their dose
            for m in ("batched", "loop", "loop_sync"):
                if m == args.mode:
                    continue
                el_o = timed(STEP[m], args.warmup, args.steps)
                other_modes[m] = {"value": round(rays_per_step * args.steps / el_o / 1e6, 2),
                                  "ms_per_step": round(el_o / args.steps * 1e3, 4), "dose_crc32": crc(rt.read_dosage())}
            other_modes["loop_sync" if args.mode != "loop_sync" else "loop"]["note"] = (
                "loop_sync = the reference's own host loop INCLUDING its clFinish after every iteration (myapp.cpp:159-165): "
                "what an unmodified MyApp::Tick sees; loop = the same calls without that sync; batched is an API extension")
            # the arithmetic + SEED semantics the reference's own kernels have on gfx950 (uvrt_set_seed_mode(1) +
            # uvrt_set_flavour(1): rays, counts and f64 maps equal the reference's live kernel chain,
            # tests/test_gpu_reference_kernels.py), same step, same mode
            if args.flavour == 0 and args.seed_mode == 0:
                rt.ctx.set_flavour(1)
                rt.ctx.set_seed_mode(1)
                el_r = timed(headline, 1, args.steps)
                crc_r = crc(rt.read_dosage())
                exp_r = golden.get("seed1_flavour1") if default_config else None
                other_modes["reference_live_chain_semantics"] = {
                    "value": round(rays_per_step * args.steps / el_r / 1e6, 2), "ms_per_step": round(el_r / args.steps * 1e3, 4),
                    "dose_crc32": crc_r, "dose_crc32_expected": exp_r, "mode": args.mode,
                    "note": "--seed-mode 1 --flavour 1: every work-item reads SEED_{k-1}, a negative seed sum converts to 0, fused "
                            "cross/dot in IntersectTri -- what the reference's cl/*.cl do on this GPU; expected CRC = the oracle's "
                            "(tests/golden/make_bench_crc.py)"}
                rt.ctx.set_flavour(0)
                rt.ctx.set_seed_mode(0)
                if exp_r is not None and crc_r != exp_r:
                    raise SystemExit("bench: seed-mode-1 / flavour-1 dose CRC %s differs from the oracle's %s" % (crc_r, exp_r))
            # ---- OPT-IN "shipped flags" arithmetic (uvrt_set_flavour 2): what the reference's own build flags compute on
            # gfx950 -- one multiply by v_rcp_f32 per slab distance instead of an exact division.  Checked against the
            # oracle in that flavour (v_rcp_f32 through the table read from this GPU) and, per ray, against the reference
            # kernel built with those flags (tests/test_gpu_shipped_flags.py)
            if args.flavour == 0 and args.seed_mode == 0 and not args.wide:
                orc_s = g.load_oracle()
                if orc_s.refgpu() is not None:
                    rt.ctx.set_flavour(2)
                    el_f = timed(headline, 1, args.steps)
                    dose_f = rt.read_dosage()
                    rt.ctx.set_flavour(0)
                    leg = {"value": round(rays_per_step * args.steps / el_f / 1e6, 2), "ms_per_step": round(el_f / args.steps * 1e3, 4),
                           "dose_crc32": crc(dose_f), "mode": args.mode,
                           "note": "--flavour 2: slab distances as (b - o) * v_rcp_f32(d), f = v_rcp_f32(a) -- the arithmetic of "
                                   "cl/extend.cl under the reference's own -cl-fast-relaxed-math build (template.cpp:1192) on this "
                                   "GPU; opt-in, the default stays flavour 0"}
                    if default_config and not args.no_cpu_baseline:
                        if oscene is None:
                            oscene = orc_s.Scene(glb)
                        ref_f = oracle_dose(oscene, orc_s.load_route(route_xml), args.waves, n_launch, 2)
                        leg["dose_crc32_expected"] = crc(ref_f)
                        leg["expected_is"] = "the oracle in flavour 2, computed in this run (v_rcp_f32 table read from this GPU)"
                        rel = np.abs(dose_f.astype(np.float64) - dose_timed) / np.maximum(np.abs(dose_timed.astype(np.float64)), 1e-30)
                        leg["triangles_beyond_1e-4_of_flavour0"] = int((rel > 1e-4).sum())
                        if leg["dose_crc32"] != leg["dose_crc32_expected"]:
                            raise SystemExit("bench: flavour-2 dose CRC %s differs from the oracle's %s" % (leg["dose_crc32"], leg["dose_crc32_expected"]))
                    other_modes["shipped_flags_flavour"] = leg
            # ---- a lamp position the context has never seen: the hot-record set-up (uvrt_hotset.hip) is inside ------
            if len(all_lamps) > 1 and not args.route:
                # five positions next to lamp 1 of the route, none of them seen before (and none a lamp of the route, whose own
                # cold figure follows below): each first computation is a cold start; the median guards against a one-off
                base = all_lamps[1]
                for _ in range(3):          # the legs above end with CPU work (the oracle): bring the GPU's clocks back up first
                    headline()
                sync_all()
                colds = []
                for k in range(1, 6):
                    configure([(base[0] + 0.015625 * k, base[1] - 0.015625 * k, base[2])], args.photons, args.waves)
                    colds.append(one_synced(headline))
                cold_ms = float(np.median(colds))
                warm_ms = median_synced(headline, few)
                cold = {"new_lamp_first_computation_ms": round(cold_ms, 4), "same_lamp_warm_ms": round(warm_ms, 4),
                        "cold_over_warm": round(cold_ms / warm_ms, 4), "new_lamp_samples_ms": [round(v, 4) for v in colds],
                        "mode": args.mode,
                        "note": "median over 5 lamp positions the context has not seen (next to lamp 1 of the route): the first "
                                "computation at each (visit statistics + hot-record selection + renumbering + per-launch records "
                                "inside), against the median of the next %d at the last of them; each bracketed by device syncs; "
                                "buffers already allocated" % few}
                configure(headline_lamps, args.photons, args.waves)
        # ---- the reference's default workload: 12 lamps x 10 iterations x 2 796 202 photons -----------------------
        if default_config and not args.lean and len(all_lamps) > 1:
            configure(all_lamps, ROUTE_PHOTONS, ROUTE_ITERATIONS)
            n_route = rt.photonsPerLight
            rays_route = ROUTE_ITERATIONS * len(all_lamps) * n_route
            # buffers of this size are allocated by a computation over SHIFTED lamps, so that the first computation of
            # the real route below is cold only in what belongs to the lamps (hot records, per-launch records)
            rt.set_lamps([(l[0] + 0.0078125, l[1] - 0.0078125, l[2]) for l in all_lamps])
            one_synced(headline)
            rt.set_lamps(all_lamps)
            cold_ms = one_synced(headline)
            crc_cold = crc(rt.read_dosage())
            k_route = max(1, min(3, args.steps))
            el_route = timed(headline, 0, k_route)
            crc_route = crc(rt.read_dosage())
            exp_route = golden.get("route_flavour%d" % args.flavour) if args.seed_mode == 0 else None
            # the same workload through the reference's own host loop with its device sync after every iteration
            # (myapp.cpp:156-170): what an unmodified MyApp::Tick sees -- 12 launches between two syncs
            el_tick = timed(step_loop_sync, 1, 1)
            crc_tick = crc(rt.read_dosage())
            route_leg = {"workload": "lange_route.xml: %d lamps x %d iterations x %d photons = %d rays per computation (raytracer.h:30-32, "
                                     "myapp.cpp:156-170)" % (len(all_lamps), ROUTE_ITERATIONS, n_route, rays_route),
                         "mode": args.mode, "ms_per_computation": round(el_route / k_route * 1e3, 3),
                         "mray_s": round(rays_route * k_route / el_route / 1e6, 1), "computations_timed": k_route,
                         "cold_first_computation_ms": round(cold_ms, 3),
                         "cold_over_warm": round(cold_ms / (el_route / k_route * 1e3), 4),
                         "unmodified_tick_loop": {"ms_per_computation": round(el_tick * 1e3, 3), "mray_s": round(rays_route / el_tick / 1e6, 1),
                                                  "dose_crc32": crc_tick,
                                                  "note": "mode loop_sync: ComputeDosageMap; Shade; sync per iteration, call by call (myapp.cpp:159-165)"},
                         "dose_crc32": crc_route, "dose_crc32_cold": crc_cold, "dose_crc32_expected": exp_route,
                         "note": "cold = the first computation over these 12 lamp positions (12 hot-record set-ups inside), bracketed "
                                 "by syncs, buffers allocated beforehand; expected CRC = the oracle's (tests/golden/make_bench_crc.py)"}
            if crc_cold != crc_route or crc_tick != crc_route or (exp_route is not None and crc_route != exp_route):
                raise SystemExit("bench: route workload dose CRC %s / %s differs from the oracle's %s" % (crc_cold, crc_route, exp_route))
            configure(headline_lamps, args.photons, args.waves)
        # timing pass for the roofline: the headline step with HIP events around the extend launches on their
        # stream (loop modes: launch pipelining off, so the kernels of neighbouring waves do not overlap)
        if args.mode != "batched":
            rt.ctx.set_pipeline(False)
        rt.ctx.set_timing(True)
        rt.ctx.extend_time_ms()                    # drop anything recorded so far
        for _ in range(timing_steps):
            headline()
        sync_all()
        ext_ms, ext_launches = rt.ctx.extend_time_ms()
        rt.ctx.set_timing(False)
        rt.ctx.set_pipeline(not args.no_pipeline)
        rt.Sync()
        # the shader clock UNDER THIS LOAD: a one-wave probe (s_memtime against the constant 100 MHz s_memrealtime) on a stream of
        # its own while the headline steps run again exactly as they were timed; the clock moves between 2.0 and 2.4 GHz with the
        # power the kernel draws, and the issue-rate peaks of the roofline move with it
        clock_steps = max(2, min(10, args.steps))
        span_us = int(0.8 * clock_steps * elapsed / args.steps * 1e6)
        if span_us >= 1:
            for _ in range(2):
                headline()
            rt.ctx.clock_probe_start(min(span_us, 1000000))
            for _ in range(clock_steps):
                headline()
            sync_all()
            clock_mhz = rt.ctx.clock_probe_read()
    else:
        # strong: BASELINE configs[3]
        rt.SetRayRange(rank, world)
        rt.set_reduce_over_comm(native_comm)
        el_s = timed(step_batched, args.warmup, args.steps)
        dose_s = rt.read_dosage()
        strong = {"value": round(rays_per_step * args.steps / el_s / 1e6, 2), "unit": "Mray/s",
                  "ms_per_step": round(el_s / args.steps * 1e3, 4), "rays_per_step": rays_per_step, "scaling": "strong",
                  "parallelism": "ray-range-sharded x%d, one int32 all-reduce of the count planes per computation (%s)"
                                 % (world, "native RCCL, uvrt_reduce_batch" if native_comm else
                                    "REHEARSAL: ranks share a GPU, gloo" if rehearsal else comm_note),
                  "dose_crc32": crc(dose_s)}
        # weak: configs[4], same run
        rt.SetRayRange(0, 1)
        rt.set_reduce_over_comm(False)
        rt.maxIterations = args.waves * world
        reducer[0] = sharding.MapReducer(rt.ctx, device)
        el_w = timed(step_weak, args.warmup, args.steps)
        dose_w = rt.read_dosage()
        weak = {"value": round(args.waves * world * n_launch * args.steps / el_w / 1e6, 2), "unit": "Mray/s",
                "ms_per_step": round(el_w / args.steps * 1e3, 4), "rays_per_step": args.waves * world * n_launch,
                "scaling": "weak", "parallelism": "launch-sharded x%d, one f64 SUM + MAX all-reduce of the maps" % world,
                "dose_crc32": crc(dose_w)}
        if args.scaling == "weak":
            elapsed, dose_timed, rays_per_step = el_w, dose_w, args.waves * world * n_launch
        else:
            elapsed, dose_timed = el_s, dose_s
    crc_timed = crc(dose_timed)
    value = rays_per_step * args.steps / elapsed / 1e6
    dose_after = rt.read_dosage()

    ranks_agree = None
    if world > 1:
        h = torch.tensor([zlib.crc32(dose_timed.tobytes())], dtype=torch.int64, device=coll_device)
        hs = [torch.zeros_like(h) for _ in range(world)]
        dist.all_gather(hs, h)
        ranks_agree = all(bool((x == hs[0]).all()) for x in hs)

    if rank == 0:
        # ---- CPU baseline + census (rank 0, N = 1 only) -----------------------------------
        cpu = None
        census = None
        expected = None
        if default_config and not (world > 1 and args.scaling == "weak"):
            expected = golden.get("seed1_flavour1" if (args.seed_mode == 1 and args.flavour == 1) else
                                  ("flavour%d" % args.flavour) if args.seed_mode == 0 else "-")
        if args.route and args.seed_mode == 0:
            expected = golden.get("route_flavour%d" % args.flavour)
        if world == 1 and not args.no_cpu_baseline:
            orc = g.load_oracle()
            if oscene is None:
                oscene = orc.Scene(glb)
            # the CPU leg always times the BASELINE workload shape (lamp 0, 8 waves of the launch size): with --route the
            # sample is 8 waves of 2 796 202 photons from lamp 0, not the whole 335 M-ray route
            cpu, census, ref_dose = cpu_baseline(oscene, orc.load_route(route_xml), WAVES if args.route else args.waves,
                                                 n_launch, args.flavour)
            if ref_dose is not None and not args.route and args.seed_mode == 0:
                same = np.array_equal(dose_timed.view(np.uint32), ref_dose.view(np.uint32))
                cpu["gpu_dose_bit_identical"] = bool(same)
                cpu["checked"] = "dose read right after the timed steps, before any other pass"
                if not same:
                    raise SystemExit("bench: the dose of the timed region differs from the oracle's")
            else:
                cpu["gpu_dose_bit_identical"] = None
                cpu["checked"] = ("the CPU sample is not the timed step (cut short, --route or --seed-mode 1): the dose is checked "
                                  "against the committed oracle CRC instead")
        if expected is not None and crc_timed != expected:
            raise SystemExit("bench: dose CRC %s of the timed region differs from the committed %s" % (crc_timed, expected))
        if census is None and os.path.exists(census_path()) and default_config:
            census = json.load(open(census_path()))["per_launch_avg"]
        kernel_name = "k_extend6<2, false, true, %d>" % args.flavour
        roof = None
        if ext_launches:
            avg_ms = ext_ms / max(ext_launches, 1)
            rays_per_extend = rays_per_step * timing_steps / max(ext_launches, 1)
            model_mode = "batched" if args.mode == "batched" else "loop"
            model_path = os.path.join(ROOT, "profiles", "extend_issue_model_%s%s.json"
                                      % (model_mode, "_flavour%d" % args.flavour if args.flavour else ""))
            step_sec = elapsed / args.steps
            if os.path.exists(model_path) and args.scene is None and not args.wide and not args.route:
                # The binding resource at the level the driver times: the whole step.  Per-ray counts x the step's rays
                # over the step's wall time -- NOT over a launch's duration, which overlaps its neighbour's on the
                # other launch lane.  The per-launch view is kept beside it, labelled.
                m = json.load(open(model_path))
                k = m["constants"]
                pr = m["per_ray"]
                clock_hz = clock_mhz * 1e6 if clock_mhz else None
                util_step, bracket_step = issue_model_utilisation(m, rays_per_step, step_sec, clock_hz)
                util_launch, bracket_launch = issue_model_utilisation(m, rays_per_extend, avg_ms * 1e-3, clock_hz)
                busiest = max(util_step, key=util_step.get)
                roof = {"bound": busiest, "kernel": kernel_name, "level": "step (driver-timed ms_per_step)",
                        "achieved": round(pr["valu_issue_cycles"] * rays_per_step / step_sec / 1e9, 1),
                        "peak": round(k["simds"] * (clock_hz or k["clock_hz_nominal"]) / 1e9, 1), "unit": "G VALU issue-cycles/s",
                        "frac": round(util_step["valu_issue"], 4),
                        "frac_bracket": [round(bracket_step[0], 4), round(bracket_step[1], 4)],
                        "clock_measured_mhz": round(clock_mhz, 1) if clock_mhz else None,
                        "clock_is": ("shader clock during %d of the timed kind of step (uvrt_clock_probe: s_memtime against the 100 MHz "
                                     "s_memrealtime, one wave on its own stream)" % clock_steps) if clock_mhz else "nominal (no probe)",
                        "frac_at_nominal_2400mhz": round(issue_model_utilisation(m, rays_per_step, step_sec, None)[0]["valu_issue"], 4),
                        "lane_utilisation": round(m["lane_utilisation"], 4),
                        "useful_lane_frac": round(util_step["valu_issue"] * m["lane_utilisation"], 4),
                        "step_level": {u: round(v, 4) for u, v in util_step.items()},
                        "traffic": round(pr["hbm_bytes"] * rays_per_extend),
                        "traffic_is": "static: PMC FETCH_SIZE x 2 + WRITE_SIZE per extend launch of %s (not collected in this run)" % m["source"],
                        "hbm_measured_frac_of_peak": round(util_step["hbm"], 4),
                        "per_launch": {"rays_per_launch": int(rays_per_extend), "avg_launch_ms": round(avg_ms, 4),
                                       "extend_mray_s": round(rays_per_extend / avg_ms / 1e3, 1),
                                       "utilisation_over_the_launch_wall_time": {u: round(v, 4) for u, v in util_launch.items()},
                                       "valu_bracket": [round(bracket_launch[0], 4), round(bracket_launch[1], 4)],
                                       "caveat": "in batched / pipelined modes two launches are co-resident: a launch's wall time "
                                                 "is not machine time, so these understate the units' load; step_level is the figure"},
                        "wave_time_waiting_on_memory": round(m["wave_wait_frac"], 3) if m.get("wave_wait_frac") else None,
                        "timing_pass": "%d step(s) after the timed region; HIP events around the extend launch on its stream" % timing_steps,
                        "model": "%s (%s)" % (os.path.relpath(model_path, ROOT), m.get("note", "")),
                        "valu_cycles_are": "per ray: stream trips x static cycles per kind of trip (%.0f %% of the VALU instructions, exact) + the "
                                           "compiler-written rest at its static mean cost +- 15 %% (tests/tools/stream_census.py)"
                                           % (100.0 * m["stream_share_of_valu_insts"]),
                        "note": "per-ray counts x rays per step / (ms_per_step x unit peak); issue peaks at the MEASURED shader clock; "
                                "deposit_atomics = scattered int32 atomic adds against the calibrated 27 G/s (every deposit leaves L2); HBM "
                                "carries a few per cent (the 5.7 MB record set lives in L2 / LDS); packed f32 saves instructions, not issue "
                                "cycles (DESIGN.md 4a)"}
            if census is not None and not args.route:
                # SURVEY.md 8d's HBM-read figure (algorithmic bytes of the REFERENCE's layout, every node visit
                # priced as a memory read): secondary on the L2-resident room, primary on a scene beyond L2
                n = float(census["rays"])
                bytes_per_ray = (32.0 + 8.0 + 32.0 * (1.0 + census["aabb_tests"] / n)
                                 + 68.0 * census["tri_tests"] / n + 4.0 * census["hits"] / n)
                achieved = bytes_per_ray * rays_per_extend / (avg_ms * 1e-3) / 1e9
                achieved_step = bytes_per_ray * rays_per_step / step_sec / 1e9
                hbm_alg = {"algorithmic_bytes_per_ray": round(bytes_per_ray, 1), "achieved_GBs": round(achieved, 1),
                           "achieved_GBs_step_level": round(achieved_step, 1),
                           "peak_GBs": HBM_PEAK_GBS, "frac": round(achieved / HBM_PEAK_GBS, 4),
                           "frac_step_level": round(achieved_step / HBM_PEAK_GBS, 4),
                           "note": "SURVEY 8d bookkeeping: every node visit priced as a memory read of the reference's "
                                   "32-B nodes / 64-B triangles; exceeds the peak where the scene is L2/LDS resident"}
                if roof is None:
                    roof = {"bound": "hbm", "bound_is": "SURVEY 8d algorithmic bytes (no per-ray issue model is committed for this scene / "
                                                        "flavour / mode; the room's figure lives in L2 / LDS, so it exceeds the peak there)",
                            "kernel": kernel_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                            "rays_per_launch": int(rays_per_extend), "avg_launch_ms": round(avg_ms, 4),
                            "extend_mray_s": round(rays_per_extend / avg_ms / 1e3, 1),
                            "note": "algorithmic bytes (SURVEY 8d) over the measured launch duration; `traffic` needs a PMC "
                                    "pass (tests/tools/pmc_extend.sh): DESIGN.md 6 has this scene's"}
                roof["hbm_algorithmic"] = hbm_alg
        if world > 1:
            step_text = ("waves x (generate, extend, accumulate, shade), whole launches dealt to ranks, one f64 SUM + MAX "
                         "all-reduce of the maps" if args.scaling == "weak" else
                         "generate + extend of this rank's ray range of all waves, ONE int32 all-reduce of the count planes, "
                         "accumulate + shade replayed per wave")
            par = (strong if args.scaling == "strong" else weak)["parallelism"]
        else:
            step_text = {"batched": "generate + extend of all waves (batched), accumulate + shade replayed per wave",
                         "loop": "waves x (generate, extend, accumulate, shade)",
                         "loop_sync": "waves x (generate, extend, accumulate, shade, device sync)"}[args.mode]
            par = "1 GPU"
        lamp_text = ("all %d lamps of lange_route.xml" % n_lamps) if args.route else "lamp 0 of lange_route.xml"
        out = {
            "metric": "Mray/s (extend+shade) on C046_1.glb 1920x1080x8-bounce", "value": round(value, 2),
            "unit": "Mray/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %d photons/launch x %d waves%s, %s, SEED_0=0; step = reset + %s + sync"
                                   % (scene_label, n_launch, args.waves, " per GPU" if (world > 1 and args.scaling == "weak") else "",
                                      lamp_text, step_text),
                       "triangles": rt.mesh.triangleCount, "rays_per_step": rays_per_step, "mode": args.mode,
                       "mode_is": ("batched = RayTracer::ComputeIterationsBatched, an API extension the reference's caller "
                                   "(myapp.cpp:156-170) does not use; other_modes.loop / loop_sync are the drop-in figures"
                                   if args.mode == "batched" else "the reference's host loop (myapp.cpp:156-165)"),
                       "launch_pipelining": bool(not args.no_pipeline), "flavour": args.flavour, "seed_mode": args.seed_mode,
                       "wide_bvh": bool(args.wide), "parallelism": par,
                       "self_comm": bool(args.self_comm), "high_priority_stream": bool(args.high_priority_stream)},
            # who reduced the count planes, and what RCCL itself says the communicator spans (ncclCommCount)
            "comm": ("native" if (native_comm or (args.self_comm and not self_torch)) else "torch" if torch_reduce else
                     "REHEARSAL (gloo; ranks share a GPU)" if rehearsal else None),
            "rccl_ranks": comm_info["rccl_ranks"] if comm_info else (dist.get_world_size() if torch_reduce else None),
            "reserved_cus": comm_info["reserved_cus"] if comm_info else 0,
            "ranks_agree": ranks_agree, "comm_note": comm_note or None,
            "roofline": roof, "cpu_baseline": cpu,
            "dose_crc32": crc_timed, "dose_crc32_expected": expected, "dose_crc32_after_all_passes": crc(dose_after),
            "value_is": "steady-state throughput of back-to-back computations (one device sync after the last step)",
        }
        if world == 1:
            same_semantics = [v["dose_crc32"] for k, v in other_modes.items()
                              if k not in ("reference_live_chain_semantics", "shipped_flags_flavour")]
            if crc(dose_after) != crc_timed or any(c != crc_timed for c in same_semantics):
                raise SystemExit("bench: the passes disagree on the dose (%s / %s / %s)" % (crc_timed, other_modes, crc(dose_after)))
            if single_ms is not None:
                out["single_computation"] = {"ms": round(single_ms, 4), "mray_s": round(rays_per_step / single_ms / 1e3, 1),
                                             "note": "one step bracketed by device syncs, median of %d" % few}
            out["other_modes"] = other_modes
            out["cold_start"] = cold
            out["route_workload"] = route_leg
        else:
            out["multi_gpu_check"] = {"dose_identical_on_all_ranks": ranks_agree, "photons_traced": rays_per_step}
            out["strong"] = strong
            out["weak"] = weak
        print(json.dumps(out), flush=True)
    if world > 1:
        with Watchdog("final barrier"):
            dist.barrier()
        if native_comm:
            rt.ctx.comm_destroy()
    if args.self_comm and not self_torch:
        rt.ctx.comm_destroy()
    if dist.is_initialized():
        dist.destroy_process_group()
    rt.close()


if __name__ == "__main__":
    main()
