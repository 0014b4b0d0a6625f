#!/bin/bash
# Developer helper for gpurun (round 3, first call): new tests first, then the whole GPU suite, the bench line and a kernel trace
OUT=gpurun_out/r3a
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_hotset.py tests/test_gpu_batch.py -m gpu -x -q > $OUT/tests_new.log 2>&1 || { tail -30 $OUT/tests_new.log; exit 1; }
tail -2 $OUT/tests_new.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || { tail -20 $OUT/bench.err; exit 1; }
tail -c 1500 $OUT/bench.json
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $REPO/$OUT/trace.log 2>&1) || echo "trace failed"
s=$(ls $OUT/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$s" ] && cp $s $OUT/kernel_stats.csv && head -20 $OUT/kernel_stats.csv
rm -rf $OUT/trace
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests_all.log 2>&1; tail -5 $OUT/tests_all.log
