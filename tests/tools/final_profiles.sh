#!/bin/bash
# Developer helper for gpurun: the profiles the round's numbers come from (bench lines, rocprofv3 kernel trace +
# stats of the default bench command, PMC passes of the extend kernel in both modes).  Output: gpurun_out/final/.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/final
mkdir -p $OUT
cd $REPO
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_batched.json 2> $OUT/bench_batched.err || echo "bench batched failed"
python3 bench.py --steps 20 --warmup 5 --mode loop > $OUT/bench_loop.json 2> $OUT/bench_loop.err || echo "bench loop failed"
python3 bench.py --steps 20 --warmup 5 --flavour 1 > $OUT/bench_flavour1.json 2> $OUT/bench_flavour1.err || echo "bench flavour 1 failed"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_batched -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lean > $OUT/trace_batched.log 2>&1) || echo "trace batched failed"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_loop -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lean --mode loop > $OUT/trace_loop.log 2>&1) || echo "trace loop failed"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_loop_nopipe -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lean --mode loop --no-pipeline > $OUT/trace_loop_nopipe.log 2>&1) || echo "trace loop nopipe failed"
for m in batched loop loop_nopipe; do
  f=$(ls $OUT/trace_$m/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 tests/tools/trace_union.py $f 10 3 $OUT/trace_${m}_union.txt
  s=$(ls $OUT/trace_$m/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$s" ] && cp $s $OUT/trace_${m}_kernel_stats.csv
done
PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --lean" bash tests/tools/pmc_extend.sh 0 0 gpurun_out/final/pmc_batched > $OUT/pmc_batched.txt 2>&1 || echo "pmc batched: a pass failed"
PIPELINE=0 bash tests/tools/pmc_extend.sh 0 0 gpurun_out/final/pmc_loop > $OUT/pmc_loop.txt 2>&1 || echo "pmc loop: a pass failed"
rm -rf $OUT/pmc_batched/p*/ $OUT/pmc_loop/p*/ $OUT/trace_*/
ls $OUT
