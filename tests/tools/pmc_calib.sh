#!/bin/bash
# Developer probe: calibrate what the SQ / TCC counters report per instruction class and per byte
# (tests/tools/valu_calib.hip, fetch_calib.hip) -- the constants behind bench.py's roofline.
# Usage on the GPU box: bash tests/tools/pmc_calib.sh [outdir]     (binaries are built here if missing)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/${1:-gpurun_out/pmc_calib}
mkdir -p $OUT
for t in valu_calib fetch_calib; do
  [ -x $REPO/tests/tools/$t ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $REPO/tests/tools/$t $REPO/tests/tools/$t.hip || exit 1
done
cd /tmp && export TMPDIR=/tmp
fail=0
run() {   # tag, counters, program, args...
  local tag=$1 ctr=$2; shift 2
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/$tag -- "$@" > $OUT/$tag.log 2>&1 || { echo "pass $tag FAILED"; fail=1; }
}
run valu_a "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" $REPO/tests/tools/valu_calib all 8
run valu_b "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" $REPO/tests/tools/valu_calib all 8
run fetch_a "FETCH_SIZE" $REPO/tests/tools/fetch_calib
run fetch_b "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum" $REPO/tests/tools/fetch_calib
python3 - <<PY
import csv, glob, collections
for tag in ("valu_a", "valu_b", "fetch_a", "fetch_b"):
    rows = collections.OrderedDict()
    for f in sorted(glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)):
        for r in csv.DictReader(open(f)):
            key = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0], r["Grid_Size"])
            rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    with open("$OUT/%s_summary.txt" % tag, "w") as o:
        for (d, k, g), c in rows.items():
            line = "%4d %-28s grid %-9s " % (d, k, g) + "  ".join("%s=%.0f" % kv for kv in sorted(c.items()))
            o.write(line + "\n")
    print(tag, len(rows), "dispatches")
PY
exit $fail
