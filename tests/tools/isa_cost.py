#!/usr/bin/env python3
"""Developer tool: price the instruction stream of a compiled kernel in VALU / SALU issue cycles.

    python tests/tools/isa_cost.py <file.s> <kernel-symbol-substring> [--blocks]

<file.s> is `hipcc -S --cuda-device-only` output.  Every instruction gets the issue cost measured by
tests/tools/valu_calib.hip on an MI355X at 8 waves per SIMD (profiles/r02/r02_valu_calibration.txt):

    2 cycles  v_fma/fmac/mul/add/sub_f32, v_mov_b32, v_and/or/xor_b32, v_lshrrev_b32, v_add/sub_u32
    4 cycles  every packed f32 op (v_pk_*), v_min/max/min3/max3/med3_f32, every v_cmp, v_cndmask,
              v_lshlrev_b32, v_lshl_add_u32, v_mul_lo/u24, v_mad_u32_u24, v_bfe/bfi/and_or/add3, 64-bit
              moves and adds, and ANY VALU instruction with an SGPR (or literal/constant-bus) source
    8 cycles  v_rcp_f32 and the other transcendentals
    SALU      1 instruction per cycle per CU, i.e. the four SIMDs of a CU share one scalar issue slot

Output: per basic block (label) the VALU instruction count, VALU issue cycles, SALU count, LDS / VMEM
counts; with --blocks the listing, otherwise the totals and the average cycles per VALU instruction.
"""
import re
import sys

FULL = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_mul_legacy_f32", "v_not_b32", "v_ashrrev_i32"}
TRANS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32",
         "v_rcp_iflag_f32", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f64"}


def base(op):
    for suf in ("_e32", "_e64", "_sdwa", "_dpp"):
        if op.endswith(suf):
            return op[: -len(suf)]
    return op


def valu_cost(op, operands):
    b = base(op)
    if b in TRANS:
        return 8
    if b.startswith("v_pk_") or b.endswith("_f64") or b.endswith("_b64") or b.endswith("_u64") or b.endswith("_i64"):
        return 4
    if b in FULL:
        # an SGPR / literal source operand halves the rate (valu_calib: v_add_f32 s, v = 4 cycles)
        srcs = operands.split(",")[1:]
        for s in srcs:
            s = s.strip().split(" ")[0]
            if re.match(r"^-?\|?s\d+|^-?\|?s\[|^vcc|^exec|^0x|^-?\d+\.\d*e|^m0", s):
                return 4
        return 2
    return 4


def main():
    path, sym = sys.argv[1], sys.argv[2]
    show = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(tuple(":")) or
                 (l.startswith("_Z") and sym in l and ":" in l.split(";")[0]))
    blocks = []
    cur = {"label": "entry", "valu": 0, "vcyc": 0, "salu": 0, "lds": 0, "vmem": 0, "other": 0, "ops": []}
    for l in lines[start + 1:]:
        t = l.split(";")[0].strip()
        if not t:
            continue
        if t.startswith(".LBB") and t.endswith(":"):
            blocks.append(cur)
            cur = {"label": t[:-1], "valu": 0, "vcyc": 0, "salu": 0, "lds": 0, "vmem": 0, "other": 0, "ops": []}
            continue
        if t.startswith(".") or t.endswith(":"):
            if t.startswith(".Lfunc_end"):
                break
            continue
        parts = t.split(None, 1)
        op, operands = parts[0], (parts[1] if len(parts) > 1 else "")
        if op.startswith("v_"):
            c = valu_cost(op, operands)
            cur["valu"] += 1
            cur["vcyc"] += c
            cur["ops"].append((op, c))
        elif op.startswith("s_"):
            if op in ("s_endpgm",):
                blocks.append(cur)
                cur = None
                break
            cur["salu"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
        else:
            cur["other"] += 1
    if cur:
        blocks.append(cur)
    tv = sum(b["valu"] for b in blocks)
    tc = sum(b["vcyc"] for b in blocks)
    ts = sum(b["salu"] for b in blocks)
    if show:
        for b in blocks:
            print("%-12s VALU %3d  cycles %4d  SALU %3d  LDS %2d  VMEM %2d" % (b["label"], b["valu"], b["vcyc"], b["salu"], b["lds"], b["vmem"]))
    print("static totals: VALU %d instructions, %d issue cycles (%.2f per instruction), SALU %d, blocks %d"
          % (tv, tc, tc / max(tv, 1), ts, len(blocks)))
    hist = {}
    for b in blocks:
        for op, c in b["ops"]:
            k = (base(op), c)
            hist[k] = hist.get(k, 0) + 1
    if show:
        for (op, c), n in sorted(hist.items(), key=lambda kv: -kv[1] * kv[0][1]):
            print("  %-22s x%3d  @%d" % (op, n, c))


if __name__ == "__main__":
    main()
