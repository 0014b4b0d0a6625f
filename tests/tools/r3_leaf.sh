#!/bin/bash
# Developer helper for gpurun (round 3): parity of the new trip stream, then the interleaved sweep of the leaf-visit rule
OUT=gpurun_out/${1:-r3b}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_hotset.py tests/test_gpu_kernels.py tests/test_gpu_adversarial.py tests/test_gpu_stress.py tests/test_gpu_batch.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
VARIANTS=${VARIANTS:-0,1201,1208,1212,1216,1308,1312,1316,1408,1412,1416,1424,1612,1812} MODE=batched ISOLATED=1 ROUNDS=${ROUNDS:-5} timeout -k 10 500 python tests/tools/ab_bench.py > $OUT/ab_batched.txt 2>&1 || { tail -5 $OUT/ab_batched.txt; exit 1; }
cat $OUT/ab_batched.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "single", d["single_computation"], "cold", d["cold_start"]["new_lamp_first_computation_ms"], d["cold_start"]["same_lamp_warm_ms"], "route", d["route_workload"]["ms_per_computation"], d["route_workload"]["cold_first_computation_ms"])
print({k: v["value"] for k, v in d["other_modes"].items()})
PY
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $REPO/$OUT/trace.log 2>&1) || echo "trace failed"
s=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$s" ] && cp $s $OUT/kernel_stats.csv && grep -E "visit_stats|select_hot|write_perm|generate_batch|extend6" $OUT/kernel_stats.csv
rm -rf $OUT/trace
