#!/bin/bash
# Developer probe: kernel trace of a rank's share of the 8-GPU step (bench.py --photons 259200) at N = 1
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/shard
mkdir -p $OUT
cd $REPO
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --photons 259200 --steps 20 --warmup 5 --no-cpu-baseline --lean > $OUT/trace.log 2>&1) || echo "trace failed"
f=$(ls $OUT/trace/*/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 tests/tools/trace_union.py $f 20 5 $OUT/union.txt && cat $OUT/union.txt
[ -n "$f" ] && python3 - $f <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# print the kernels of two consecutive steps in the middle
mid=len(rows)//2
t0=int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid+24]:
    print("%9.1f us  +%8.1f us  %-40s q%s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:40], r.get("Queue_Id","?")))
PY
rm -rf $OUT/trace
