#!/usr/bin/env python3
"""Developer tool: price the VALU issue cycles of k_extend6 per ray EXACTLY where the instruction stream is hand-written, and
within a narrow bracket where hipcc wrote it.

The product's trips run in ONE hand-written asm statement (uvrt_extend6.hip R7_BODY): its instructions per kind of trip are
static, so   cycles(stream) = sum over trip kinds of  count(kind) x cycles(kind)
with the counts from a -DUVRT_TRIP_STATS build of the same loop (the "trip census:" line of uvrt_sync; tests/tools/trip_census.sh)
and the cycles per instruction from the calibration (tests/tools/valu_calib.hip, profiles/r02/r02_valu_calibration.txt: 2 / 4 / 8
cycles per wave64 instruction, isa_cost.valu_cost).  What is left -- refills, general-step trips, prologue: hipcc's code -- is
(PMC SQ_INSTS_VALU - the stream's instruction count) instructions, priced with the static mean cost of the kernel's
compiler-written VALU instructions (hipcc -S, everything outside #APP ... #NO_APP) +- 15 %.

    python tests/tools/stream_census.py --flavour 0 --census gpurun_out/<tag>/census_f0.txt --pmc profiles/.../summary.txt \
           --rays-per-launch 5529600 --out profiles/extend_issue_model_batched.json
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import isa_cost  # noqa: E402

SRC = os.path.join(ROOT, "small-project-uv-robot-ray-tracer_amd", "csrc", "uvrt_extend6.hip")
FLAVOUR_BODY = {
    0: "R7_BODY(R7_TRI(R7_CROSS_STRICT, R7_DOT_STRICT, R7_NEWTON_EXACT, R7_GO_A_STRICT, R7_GO_01_STRICT), R7_SLABS_EXACT)",
    1: "R7_BODY(R7_TRI(R7_CROSS_OCL, R7_DOT_OCL, R7_NEWTON_EXACT, R7_GO_A_STRICT, R7_GO_01_STRICT), R7_SLABS_EXACT)",
    2: "R7_BODY(R7_TRI(R7_CROSS_OCL, R7_DOT_OCL, R7_NEWTON_NONE, R7_GO_A_SHIPPED, R7_GO_01_SHIPPED), R7_SLABS_SHIPPED)",
}
# operand kinds of the asm statement (R7_OPERANDS): scalar registers / everything else is a vector register
SCALAR = {"km", "code", "m0", "m1", "m2", "m3", "m4", "tb", "full", "rb", "spec", "tp", "amin", "ox", "oz"}


def stream_text(flavour):
    """the asm string of run7<flavour>: the macro block of uvrt_extend6.hip through the C preprocessor"""
    lines = open(SRC).read().split("\n")
    a = next(i for i, l in enumerate(lines) if l.startswith("#define R7_CROSS_STRICT"))
    b = next(i for i, l in enumerate(lines) if l.startswith("#define R7_OPERANDS"))
    with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as f:
        f.write("\n".join(lines[a:b]) + "\nCENSUS_BEGIN " + FLAVOUR_BODY[flavour] + " CENSUS_END\n")
        path = f.name
    out = subprocess.check_output(["gcc", "-E", "-P", path], text=True)
    os.unlink(path)
    body = out[out.index("CENSUS_BEGIN") + 12:out.index("CENSUS_END")]
    text = "".join(bytes(m, "utf-8").decode("unicode_escape") for m in re.findall(r'"((?:[^"\\]|\\.)*)"', body))
    return [l.strip() for l in text.split("\n") if l.strip()]


def classify(ins):
    """-> (unit, cycles): unit in valu / salu / lds / vmem / label / other"""
    if ins.endswith(":"):
        return "label", 0
    parts = ins.split(None, 1)
    op, operands = parts[0], (parts[1] if len(parts) > 1 else "")
    operands = re.sub(r"%\[(\w+)\]", lambda m: ("s90" if m.group(1) in SCALAR else "v90"), operands)
    if op.startswith("v_"):
        return "valu", isa_cost.valu_cost(op, operands)
    if op.startswith("s_"):
        return "salu", 1
    if op.startswith("ds_"):
        return "lds", 0
    if op.startswith("global_"):
        return "vmem", 0
    return "other", 0


def segments(ins):
    """split the stream at its control-flow points (uvrt_extend6.hip R7_BODY): A entry checks, B leaf-code check, C fetch,
    D triangle block, E, F box block + descend, G pop-only tail"""
    def find(pred, start=0):
        return next(i for i in range(start, len(ins)) if pred(ins[i]))
    i1 = find(lambda s: s == "1:")
    iB = find(lambda s: s.startswith("s_cbranch_scc0 3f")) + 1
    i3 = find(lambda s: s == "3:")
    iD = find(lambda s: s.startswith("s_cbranch_scc1 4f")) + 1
    i4 = find(lambda s: s == "4:")
    iF = find(lambda s: s.startswith("s_cbranch_scc1 5f")) + 1
    i5 = find(lambda s: s == "5:")
    i7 = find(lambda s: s == "7:")
    return {"A": ins[i1:iB], "B": ins[iB:i3], "C": ins[i3:iD], "D": ins[iD:i4], "E": ins[i4:iF], "F": ins[iF:i5], "G": ins[i5:i7]}


def tally(seg):
    t = {"valu": 0, "valu_cycles": 0, "salu": 0, "lds": 0, "vmem": 0, "by_cost": {2: 0, 4: 0, 8: 0}}
    for s in seg:
        u, c = classify(s)
        if u == "valu":
            t["valu"] += 1
            t["valu_cycles"] += c
            t["by_cost"][c] += 1
        elif u in ("salu", "lds", "vmem"):
            t[u] += 1
    return t


def compiler_written_mean_cost(flavour):
    """static mean issue cost of the VALU instructions hipcc wrote in k_extend6<2, false, true, flavour> (everything outside the
    inline asm statements), from hipcc -S"""
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
             "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-S", "--cuda-device-only"]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + [SRC, "-o", out], stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    sym = "_ZN4uvrt9k_extend6ILi2ELb0ELb1ELi%dEEEvNS_12ExtendParamsE" % flavour
    a = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
    b = next(i for i in range(a, len(lines)) if lines[i].lstrip().startswith("s_endpgm"))
    inside, n, cyc, hist = False, 0, 0, {2: 0, 4: 0, 8: 0}
    for l in lines[a:b]:
        t = l.split(";")[0].strip() if not l.lstrip().startswith(";") else l.strip()
        if l.lstrip().startswith(";APP") or l.lstrip().startswith("#APP"):
            inside = True
            continue
        if l.lstrip().startswith(";NO_APP") or l.lstrip().startswith("#NO_APP"):
            inside = False
            continue
        if inside or not t.startswith("v_"):
            continue
        parts = t.split(None, 1)
        c = isa_cost.valu_cost(parts[0], parts[1] if len(parts) > 1 else "")
        n += 1
        cyc += c
        hist[c] = hist.get(c, 0) + 1
    return n, cyc / max(n, 1), hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flavour", type=int, default=0)
    ap.add_argument("--census", help="file with the 'trip census:' lines of a -DUVRT_TRIP_STATS run and a 'census rays N' line")
    ap.add_argument("--pmc", help="PMC summary (tests/tools/pmc_extend.sh) of the same kernel and workload")
    ap.add_argument("--rays-per-launch", type=float, default=5529600.0)
    ap.add_argument("--out")
    ap.add_argument("--note", default="")
    ap.add_argument("--show", action="store_true")
    args = ap.parse_args()

    ins = stream_text(args.flavour)
    seg = {k: tally(v) for k, v in segments(ins).items()}
    kinds = {"stream_in": "ACEF", "stream_leaf": "ABCDEG", "stream_both": "ABCDEF", "exit": "A"}
    per_kind = {}
    for k, letters in kinds.items():
        per_kind[k] = {f: sum(seg[s][f] for s in letters) for f in ("valu", "valu_cycles", "salu", "lds", "vmem")}
        per_kind[k]["by_cost"] = {c: sum(seg[s]["by_cost"][c] for s in letters) for c in (2, 4, 8)}
    if args.show or not args.census:
        for k, v in seg.items():
            print("segment %s: %s" % (k, v))
        for k, v in per_kind.items():
            print("trip kind %-12s VALU %3d instructions = %3d issue cycles (%.2f per instruction)  SALU %2d  LDS %d  VMEM %d"
                  % (k, v["valu"], v["valu_cycles"], v["valu_cycles"] / max(v["valu"], 1), v["salu"], v["lds"], v["vmem"]))
    if not args.census:
        return
    tot = {}
    rays = 0
    for l in open(args.census):
        m = re.search(r"trip census: (.*)", l)
        if m:
            kv = m.group(1).split()
            for k, v in zip(kv[0::2], kv[1::2]):
                tot[k] = tot.get(k, 0) + int(v)
        m = re.search(r"census rays (\d+)", l)
        if m:
            rays += int(m.group(1))
    assert rays > 0 and tot, "no census in %s" % args.census
    n = {"stream_in": tot["stream_in"], "stream_leaf": tot["stream_leaf"], "stream_both": tot["stream_both"],
         "exit": tot["refills"] + tot["general_exact"] + tot["general_other"] + tot["waves"]}
    stream = {f: sum(n[k] * per_kind[k][f] for k in n) / rays for f in ("valu", "valu_cycles", "salu", "lds", "vmem")}
    pmc = {}
    for l in open(args.pmc):
        m = re.match(r"(\S+)\s+per-launch avg\s+([0-9.]+)", l)
        if m:
            pmc[m.group(1)] = float(m.group(2)) / args.rays_per_launch
    rest_valu = pmc["SQ_INSTS_VALU"] - stream["valu"]
    rest_salu = pmc["SQ_INSTS_SALU"] - stream["salu"]
    n_cw, mean_cw, hist_cw = compiler_written_mean_cost(args.flavour)
    est = stream["valu_cycles"] + rest_valu * mean_cw
    lo = stream["valu_cycles"] + rest_valu * mean_cw * 0.85
    hi = stream["valu_cycles"] + rest_valu * mean_cw * 1.15
    fetch_kb, write_kb = pmc.get("FETCH_SIZE", 0.0), pmc.get("WRITE_SIZE", 0.0)
    model = {
        "source": os.path.relpath(args.pmc, ROOT), "census": os.path.relpath(args.census, ROOT), "note": args.note,
        "flavour": args.flavour, "rays_per_launch": args.rays_per_launch,
        "method": "stream trips: count x static cycles per kind (exact); the rest (refills, general steps, prologue: hipcc's code) = "
                  "(PMC VALU instructions - stream instructions) x the static mean cost of the compiler-written instructions +- 15 %",
        "trips_per_ray": {k: n[k] / rays for k in n}, "general_trips_per_ray": (tot["general_exact"] + tot["general_other"]) / rays,
        "refills_per_ray": tot["refills"] / rays, "leaf_lane_tests_per_ray": tot["leaf_lane_tests"] / rays,
        "per_trip_kind": per_kind,
        "stream_per_ray": stream,
        "compiler_written": {"valu_insts_per_ray": rest_valu, "salu_insts_per_ray": rest_salu, "static_instructions": n_cw,
                             "static_mean_cycles": mean_cw, "static_by_cost": hist_cw},
        "valu_insts_pmc_per_ray": pmc["SQ_INSTS_VALU"], "stream_share_of_valu_insts": stream["valu"] / pmc["SQ_INSTS_VALU"],
        "valu_issue_cycles": {"estimate": est * args.rays_per_launch, "lower": lo * args.rays_per_launch, "upper": hi * args.rays_per_launch},
        "lane_utilisation": pmc.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * pmc["SQ_INSTS_VALU"]),
        "salu_insts": pmc["SQ_INSTS_SALU"] * args.rays_per_launch,
        "wave_wait_frac": pmc.get("SQ_WAIT_ANY", 0.0) / pmc["SQ_WAVE_CYCLES"] if pmc.get("SQ_WAVE_CYCLES") else None,
        "l2_hit_rate": (pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"])) if "TCC_HIT_sum" in pmc else None,
        "hbm_bytes": (2.0 * fetch_kb + write_kb) * 1024 * args.rays_per_launch,
        "per_ray": {"valu_insts": pmc["SQ_INSTS_VALU"], "valu_issue_cycles": est, "valu_issue_cycles_bracket": [lo, hi],
                    "salu_insts": pmc["SQ_INSTS_SALU"], "l1_lane_lookups": pmc.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0),
                    "hbm_bytes": (2.0 * fetch_kb + write_kb) * 1024, "deposit_atomics": pmc.get("TCC_EA0_ATOMIC_sum", 0.0)},
        # unit peaks: issue slots scale with the shader clock, which bench.py MEASURES during its steps (uvrt_clock_probe_*);
        # calibrations: profiles/r02/r02_valu_calibration.txt, r02_l1_lookup_calibration.txt, profiles/r04/r04_atomic_calibration.txt
        "constants": {"simds": 1024, "cus": 256, "clock_hz_nominal": 2.4e9, "l1_lookups_per_clk_per_cu": 1.7,
                      "hbm_peak_bytes_per_s": 8.0e12, "scattered_atomic_adds_per_s": 27.0e9},
    }
    print(json.dumps({"stream_per_ray": stream, "rest_valu_per_ray": rest_valu, "mean_cost_rest": mean_cw,
                      "valu_issue_cycles_per_ray": [lo, est, hi], "bracket_rel": [(lo / est - 1), (hi / est - 1)],
                      "stream_share": model["stream_share_of_valu_insts"]}, indent=1))
    if args.out:
        json.dump(model, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
