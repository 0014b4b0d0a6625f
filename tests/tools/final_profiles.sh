#!/bin/bash
# Developer helper for gpurun: the profiles the round's numbers come from (bench lines, rocprofv3 kernel trace + stats of the
# default bench command, PMC passes of the extend kernel per mode / flavour, the trip census).  Output: gpurun_out/<tag>/.
#   bash tests/tools/final_profiles.sh <tag> [nopmc]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-final}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "bench default failed"
python3 bench.py --steps 20 --warmup 5 --mode loop --no-cpu-baseline --lean > $OUT/bench_loop.json 2> $OUT/bench_loop.err || echo "bench loop failed"
python3 bench.py --steps 20 --warmup 5 --mode loop_sync --no-cpu-baseline --lean > $OUT/bench_loop_sync.json 2> $OUT/bench_loop_sync.err || echo "bench loop_sync failed"
python3 bench.py --steps 20 --warmup 5 --flavour 2 --lean > $OUT/bench_flavour2.json 2> $OUT/bench_flavour2.err || echo "bench flavour 2 failed"
python3 bench.py --steps 20 --warmup 5 --flavour 2 --mode loop_sync --no-cpu-baseline --lean > $OUT/bench_flavour2_loop_sync.json 2> $OUT/bench_flavour2_loop_sync.err || echo "bench flavour 2 loop_sync failed"
python3 bench.py --steps 20 --warmup 5 --flavour 1 --seed-mode 1 --no-cpu-baseline --lean > $OUT/bench_reference_semantics.json 2> $OUT/bench_reference_semantics.err || echo "bench flavour 1 / seed mode 1 failed"
python3 bench.py --steps 5 --warmup 2 --route > $OUT/bench_route.json 2> $OUT/bench_route.err || echo "bench route failed"
python3 bench.py --steps 3 --warmup 1 --route --mode loop_sync --no-cpu-baseline --lean > $OUT/bench_route_loop_sync.json 2> $OUT/bench_route_loop_sync.err || echo "bench route loop_sync failed"
python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err || echo "bench --gpus 2 (rehearsal) failed"
# the default command under the kernel trace (every leg: the per-lamp set-up kernels show here), then the lean traces per mode
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace_default.log 2>&1) || echo "trace default failed"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_batched -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lean > $OUT/trace_batched.log 2>&1) || echo "trace batched failed"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_loop_sync -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lean --mode loop_sync > $OUT/trace_loop_sync.log 2>&1) || echo "trace loop_sync failed"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_loop_nopipe -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --lean --mode loop --no-pipeline > $OUT/trace_loop_nopipe.log 2>&1) || echo "trace loop nopipe failed"
for m in default batched loop_sync loop_nopipe; do
  f=$(ls $OUT/trace_$m/*/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && [ $m != default ] && python3 tests/tools/trace_union.py $f 10 3 $OUT/trace_${m}_union.txt
  s=$(ls $OUT/trace_$m/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$s" ] && cp $s $OUT/trace_${m}_kernel_stats.csv
done
if [ "${2:-}" != nopmc ]; then
  PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --lean" bash tests/tools/pmc_extend.sh 0 0 gpurun_out/$TAG/pmc_batched > $OUT/pmc_batched.txt 2>&1 || echo "pmc batched: a pass failed"
  PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --lean --flavour 2" bash tests/tools/pmc_extend.sh 0 0 gpurun_out/$TAG/pmc_batched_flavour2 > $OUT/pmc_batched_flavour2.txt 2>&1 || echo "pmc batched flavour 2: a pass failed"
  PIPELINE=0 bash tests/tools/pmc_extend.sh 0 0 gpurun_out/$TAG/pmc_loop > $OUT/pmc_loop.txt 2>&1 || echo "pmc loop: a pass failed"
  PIPELINE=0 bash tests/tools/pmc_extend.sh 0 0 gpurun_out/$TAG/pmc_loop_flavour2 2 > $OUT/pmc_loop_flavour2.txt 2>&1 || echo "pmc loop flavour 2: a pass failed"
fi
rm -rf $OUT/pmc_batched/p*/ $OUT/pmc_batched_flavour2/p*/ $OUT/pmc_loop/p*/ $OUT/pmc_loop_flavour2/p*/ $OUT/trace_*/
bash tests/tools/trip_census.sh $TAG > $OUT/trip_census.log 2>&1 || echo "trip census failed"
ls $OUT
