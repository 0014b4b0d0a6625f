#!/bin/bash
# Developer probe: kernel trace of a rank's share of the 8-GPU step (bench.py --photons 259200 --self-comm) at N = 1: the
# step launches the REAL all-reduce kernel of the count planes (a one-rank RCCL communicator) between tracing and replay.
# Where does the collective land relative to the persistent extend waves, and how long does it take?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-shard}
mkdir -p $OUT
cd $REPO
for tag in default prio; do
  extra=""; [ $tag = prio ] && extra="--high-priority-stream"
  timeout -k 10 300 python3 bench.py --photons 259200 --steps 40 --warmup 5 --no-cpu-baseline --lean --self-comm $extra > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err || echo "bench $tag failed"
  timeout -k 10 300 python3 bench.py --photons 259200 --steps 40 --warmup 5 --no-cpu-baseline --lean $extra > $OUT/bench_${tag}_nocomm.json 2> $OUT/bench_${tag}_nocomm.err || echo "bench $tag nocomm failed"
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $REPO/bench.py --photons 259200 --steps 20 --warmup 5 --no-cpu-baseline --lean --self-comm $extra > $OUT/trace_$tag.log 2>&1) || echo "trace $tag failed"
  f=$(ls $OUT/trace_$tag/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tests/tools/trace_union.py $f 20 5 $OUT/union_$tag.txt && cat $OUT/union_$tag.txt
  [ -n "$f" ] && python3 - $f > $OUT/timeline_$tag.txt <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the kernels of two consecutive steps in the middle of the timed region
mid=len(rows)//2
t0=int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid+26]:
    print("%9.1f us  +%8.1f us  %-44s q%s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:44], r.get("Queue_Id","?")))
# the collective's delay: from the end of the fold that feeds it to its own start, and its duration
prev=None; waits=[]; durs=[]
for r in rows:
    n=r["Kernel_Name"]
    if "k_fold_planes" in n: prev=int(r["End_Timestamp"])
    elif ("nccl" in n.lower() or "AllReduce" in n) and prev is not None:
        waits.append((int(r["Start_Timestamp"])-prev)/1e3); durs.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3); prev=None
if not waits:
    print("no RCCL kernel in the trace (a one-rank communicator launches none for an in-place all-reduce)")
else:
    waits.sort(); durs.sort()
    print("all-reduce kernel: %d launches; start after its fold: median %.1f us, p90 %.1f us; duration median %.1f us, p90 %.1f us" % (len(waits), waits[len(waits)//2], waits[int(len(waits)*0.9)], durs[len(durs)//2], durs[int(len(durs)*0.9)]))
PY
  [ -f $OUT/timeline_$tag.txt ] && cat $OUT/timeline_$tag.txt
  rm -rf $OUT/trace_$tag
done
python3 - <<PY
import json
for tag in ("default", "default_nocomm", "prio", "prio_nocomm"):
    try:
        d=json.loads(open("$OUT/bench_%s.json" % tag).read().strip().splitlines()[-1]); print(tag, d["value"], d["ms_per_step"], d["dose_crc32"])
    except Exception as e: print(tag, "no line", e)
PY
