#!/usr/bin/env python3
"""Developer tool: turn a PMC summary of the extend kernel (tests/tools/pmc_extend.sh) and the issue-rate
calibration (tests/tools/valu_calib.hip, profiles/r02/r02_valu_calibration.txt) into the per-launch resource
model bench.py's `roofline` prices a launch with.

    python tests/tools/issue_model.py <summary.txt> <out.json> [rays_per_launch] [note]

Calibrated constants (MI355X, 8 waves per SIMD, profiles/r02/r02_*calibration*.txt):
  * VALU issue, cycles per wave64 instruction per SIMD: 2 for v_fma/mul/add/sub_f32, v_mov, v_and/or/xor,
    v_add/sub_u32, v_lshrrev; 4 for every packed f32 op, v_min/max/min3/max3, every v_cmp, v_cndmask, shifts
    left, 24/32-bit multiplies, three-operand integer ops and any VALU op with an SGPR source; 8 for v_rcp_f32.
    Packed f32 therefore saves instructions, not issue cycles (2 results per 4 cycles).
  * SQ_ACTIVE_INST_VALU counts ONE unit per instruction for 2- and 4-cycle classes alike (2 for v_rcp_f32):
    it is an instruction count in disguise, not a cycle count; SQ_THREAD_CYCLES_VALU = active lanes summed
    over instructions.
  * SALU: one instruction per cycle per CU (the four SIMDs share the scalar unit); it issues beside the VALU.
  * L1 (TCP): 1.6-1.8 lane-lookups per clock per CU when every lookup hits (16-byte loads, random slots).
  * FETCH_SIZE counts 64 B per 128-byte fabric read: bytes = 2 x FETCH_SIZE (stream and gather alike).
The kernel's instruction stream per launch is deterministic (static ray ownership per wave), so the
counts below are properties of (kernel build, scene, lamp, ray count), not of a particular run.
"""
import json
import re
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    rays = int(sys.argv[3]) if len(sys.argv) > 3 else 2073600
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    c = {}
    for line in open(src):
        m = re.match(r"(\S+)\s+per-launch avg\s+([0-9.]+)", line)
        if m:
            c[m.group(1)] = float(m.group(2))
    valu = c["SQ_INSTS_VALU"]
    trans = c.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
    fma, mul, add = c.get("SQ_INSTS_VALU_FMA_F32", 0.0), c.get("SQ_INSTS_VALU_MUL_F32", 0.0), c.get("SQ_INSTS_VALU_ADD_F32", 0.0)
    int32, cvt = c.get("SQ_INSTS_VALU_INT32", 0.0), c.get("SQ_INSTS_VALU_CVT", 0.0)
    # inner-block executions: 12 v_pk_fma_f32 each; every v_rcp_f32 brings two scalar fma (Newton step)
    e_in = max(0.0, (fma - 2.0 * trans) / 12.0)
    packed = 20.0 * e_in
    scalar_f32 = max(0.0, (fma - 12.0 * e_in) + (mul - 6.0 * e_in) + (add - 2.0 * e_in))
    other = max(0.0, valu - packed - trans - scalar_f32 - int32 - cvt)
    lo = packed * 4 + trans * 8 + scalar_f32 * 2 + int32 * 2 + cvt * 4 + other * 2
    hi = packed * 4 + trans * 8 + scalar_f32 * 4 + int32 * 4 + cvt * 4 + other * 4
    # central estimate: scalar f32 at 2 (a few have an SGPR operand), integer at 3 (compares and shifts at 4,
    # adds at 2), "other" = compares / min / max / selects at 4 except the register moves (about a third) at 2
    mid = packed * 4 + trans * 8 + scalar_f32 * 2.2 + int32 * 3 + cvt * 4 + other * (4 * 2.0 / 3 + 2 * 1.0 / 3)
    fetch_kb, write_kb = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
    model = {
        "source": src, "note": note, "rays_per_launch": rays,
        "valu_insts": valu, "valu_classes": {"packed_f32": packed, "trans_f32": trans, "scalar_f32": scalar_f32,
                                              "int32": int32, "cvt": cvt, "other(cmp,min,max,select,mov)": other},
        "inner_block_executions": e_in,
        "valu_issue_cycles": {"estimate": mid, "lower": lo, "upper": hi},
        "lane_utilisation": c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * valu) if valu else None,
        "salu_insts": c.get("SQ_INSTS_SALU", 0.0),
        "l1_lane_lookups": c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0),
        "l1_misses_to_l2": c.get("TCP_TCC_READ_REQ_sum", 0.0),
        "l2_hit_rate": (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])) if "TCC_HIT_sum" in c else None,
        "lds_insts": c.get("SQ_INSTS_LDS", 0.0), "vmem_rd_insts": c.get("SQ_INSTS_VMEM_RD", 0.0),
        "wave_wait_frac": c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None,
        "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
        "hbm_bytes": 2.0 * fetch_kb * 1024 + write_kb * 1024,
        "per_ray": {"valu_insts": valu / rays, "valu_issue_cycles": mid / rays, "salu_insts": c.get("SQ_INSTS_SALU", 0.0) / rays,
                    "l1_lane_lookups": c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / rays,
                    "hbm_bytes": (2.0 * fetch_kb * 1024 + write_kb * 1024) / rays},
        "constants": {"simds": 1024, "cus": 256, "clock_hz": 2.4e9, "l1_lookups_per_clk_per_cu": 1.7,
                      "hbm_peak_bytes_per_s": 8.0e12},
    }
    json.dump(model, open(out, "w"), indent=1)
    print(json.dumps(model["per_ray"]), "valu cycles est/lo/hi %.0fM/%.0fM/%.0fM" % (mid / 1e6, lo / 1e6, hi / 1e6))


if __name__ == "__main__":
    main()
