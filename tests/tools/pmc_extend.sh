#!/bin/bash
# Developer probe: rocprofv3 PMC passes over the extend kernel (one variant / sort setting per
# call).  Usage on the GPU box:  bash tests/tools/pmc_extend.sh <variant> <sort_bits> <outdir> [flavour]
# Counters are collected in separate passes (--pmc only, no tracing domains); a pass that fails makes
# the script fail (exit 1) after the remaining passes have run.
set -u
V=${1:-0}; S=${2:-0}; OUT=${3:-gpurun_out/pmc_v$V}; F=${4:-0}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/$OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"
 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_SCA"
 "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH"
 "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_LEVEL_VMEM"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
 "TA_FLAT_READ_WAVEFRONTS_sum TCP_TA_TCP_STATE_READ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum"
)
# PMC_ARGS: profile `python3 bench.py $PMC_ARGS` instead of the quick probe (e.g. "--scene soup:1000000 --steps 3
# --warmup 1 --no-cpu-baseline --no-pipeline"); PMC_PASSES: only the passes whose counters match this regex
i=0
fail=0
for P in "${PASSES[@]}"; do
  if [ -n "${PMC_PASSES:-}" ] && ! echo "$P" | grep -Eq "$PMC_PASSES"; then i=$((i+1)); continue; fi
  if [ -n "${PMC_ARGS:-}" ]; then
    (cd $REPO && timeout -k 10 400 rocprofv3 --pmc $P --output-format csv -d $REPO/$OUT/p$i -- python3 $REPO/bench.py $PMC_ARGS > $REPO/$OUT/p$i.log 2>&1) || { echo "pass $i ($P) FAILED"; fail=1; }
  else
    N=${N:-2073600} VARIANTS=$V SORTS=$S FLAVOUR=$F CHECK=0 timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $REPO/$OUT/p$i -- python3 $REPO/tests/tools/quick_extend_bench.py > $REPO/$OUT/p$i.log 2>&1 || { echo "pass $i ($P) FAILED"; fail=1; }
  fi
  i=$((i+1))
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$REPO/$OUT/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_extend" not in row["Kernel_Name"]:
            continue
        a = agg[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
with open("$REPO/$OUT/summary.txt", "w") as o:
    for k in sorted(agg):
        line = "%-40s per-launch avg %16.1f  (launches %d)" % (k, agg[k][0] / agg[k][1], agg[k][1])
        print(line); o.write(line + "\n")
PY
exit $fail
