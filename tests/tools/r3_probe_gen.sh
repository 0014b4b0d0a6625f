#!/bin/bash
# Developer probe (gpurun): what k_generate_batch costs the batched step -- the developer build skips the generate launches
# after the warm-up (UVRT_PROBE_SKIP_GENERATE; the buffers still hold the same rays, the dose stays right), alternating
# processes with and without the skip.  An upper bound for any scheme that makes the rays inside the traversal kernel.
OUT=gpurun_out/${1:-r3c}
mkdir -p $OUT
P=small-project-uv-robot-ray-tracer_amd
timeout -k 10 300 python -m pytest tests/test_gpu_hotset.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
cp $P/libuvrt_hip.so /tmp/libuvrt_keep.so
cp $P/libuvrt_hip_dev.so $P/libuvrt_hip.so
for i in 1 2 3; do
  for skip in 0 4; do
    UVRT_PROBE_SKIP_GENERATE=$skip VARIANTS=0 MODE=batched ROUNDS=4 STEPS=30 timeout -k 10 200 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/skip_generate=$skip  /"
  done
done | tee $OUT/probe_skip_generate.txt
cp /tmp/libuvrt_keep.so $P/libuvrt_hip.so
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
