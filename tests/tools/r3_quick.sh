#!/bin/bash
# Developer helper for gpurun: hot-record tests, the bench line (cold / route legs) and the kernel statistics of the default bench command
OUT=gpurun_out/${1:-r3q}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_hotset.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "single", d["single_computation"]["ms"], "cold", d["cold_start"]["new_lamp_first_computation_ms"], d["cold_start"]["same_lamp_warm_ms"], "route", d["route_workload"]["ms_per_computation"], d["route_workload"]["cold_first_computation_ms"])
PY
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $REPO/$OUT/trace.log 2>&1) || echo "trace failed"
s=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$s" ] && cp $s $OUT/kernel_stats.csv && grep -E "visit_stats|select_hot|write_perm|generate_batch|extend6" $OUT/kernel_stats.csv
rm -rf $OUT/trace
