#!/bin/bash
# Developer probe: memory-side PMC passes over the extend kernel.
# Usage on the GPU box:  bash tests/tools/pmc_mem.sh <variant> <sort_bits> <outdir>
set -u
V=${1:-0}; S=${2:-0}; OUT=${3:-gpurun_out/pmcm_v$V}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $REPO/$OUT
cd /tmp && export TMPDIR=/tmp
PASSES=(
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
 "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TOTAL_CACHE_ACCESSES_sum"
 "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_REQ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_ATOMIC_sum"
 "TA_TA_BUSY_sum TD_TD_BUSY_sum TD_TC_STALL_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
 "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
)
i=0
for P in "${PASSES[@]}"; do
  N=${N:-2073600} VARIANTS=$V SORTS=$S CHECK=0 timeout -k 10 200 rocprofv3 --pmc $P --output-format csv -d $REPO/$OUT/p$i -- python3 $REPO/tests/tools/quick_extend_bench.py > $REPO/$OUT/p$i.log 2>&1 || echo "pass $i failed"
  i=$((i+1))
done
grep -h "variant" $REPO/$OUT/p7.log
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
