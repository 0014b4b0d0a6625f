#!/bin/bash
# round 4: the drop-in figure (loop_sync = the unmodified MyApp::Tick): the whole suite, then A/B of the next-iteration hint
TAG=${1:-r4d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1; echo "suite rc=$?"; tail -6 $OUT/tests.log
for mode in loop_sync loop batched; do
  FLAVOURS=0,2 AHEADS=1,0 VARIANTS=0 MODE=$mode ROUNDS=5 STEPS=20 timeout -k 10 300 python tests/tools/ab_bench.py 2>&1 | grep "^variant\|MISMATCH\|rror"
done | tee $OUT/ab_dropin.txt
LAMPS=12 PHOTONS=2796202 WAVES=10 FLAVOURS=0 AHEADS=1,0 VARIANTS=0 MODE=loop_sync ROUNDS=3 STEPS=2 timeout -k 10 300 python tests/tools/ab_bench.py 2>&1 | grep "^variant\|MISMATCH\|rror" | sed 's/^/route: /' | tee -a $OUT/ab_dropin.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace_loop_sync -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lean --mode loop_sync > $GRAFT_REPO_ROOT/$OUT/trace_loop_sync.log 2>&1) || echo "trace failed"
s=$(ls $OUT/trace_loop_sync/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$s" ] && cp $s $OUT/trace_loop_sync_kernel_stats.csv && head -8 $OUT/trace_loop_sync_kernel_stats.csv | cut -c1-160
f=$(ls $OUT/trace_loop_sync/*/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python3 tests/tools/trace_union.py $f 10 3 $OUT/trace_loop_sync_union.txt && head -12 $OUT/trace_loop_sync_union.txt
rm -rf $OUT/trace_loop_sync
