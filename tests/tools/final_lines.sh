#!/bin/bash
# Developer helper for gpurun: the kept bench lines (second pass of final_profiles.sh, after tests/tools/stream_census.py has turned
# the first pass's PMC summaries and trip census into profiles/extend_issue_model_*.json), + the flavour-1 counters and census.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-lines}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd $REPO
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "bench default failed"
python3 bench.py --steps 20 --warmup 5 --mode loop --no-cpu-baseline --lean > $OUT/bench_loop.json 2> $OUT/bench_loop.err || echo "bench loop failed"
python3 bench.py --steps 20 --warmup 5 --mode loop_sync --no-cpu-baseline --lean > $OUT/bench_loop_sync.json 2> $OUT/bench_loop_sync.err || echo "bench loop_sync failed"
python3 bench.py --steps 20 --warmup 5 --flavour 2 --lean > $OUT/bench_flavour2.json 2> $OUT/bench_flavour2.err || echo "bench flavour 2 failed"
python3 bench.py --steps 20 --warmup 5 --flavour 2 --mode loop_sync --no-cpu-baseline --lean > $OUT/bench_flavour2_loop_sync.json 2> $OUT/bench_flavour2_loop_sync.err || echo "bench flavour 2 loop_sync failed"
python3 bench.py --steps 5 --warmup 2 --route > $OUT/bench_route.json 2> $OUT/bench_route.err || echo "bench route failed"
python3 bench.py --steps 3 --warmup 1 --route --mode loop_sync --no-cpu-baseline --lean > $OUT/bench_route_loop_sync.json 2> $OUT/bench_route_loop_sync.err || echo "bench route loop_sync failed"
python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_gpus2_rehearsal.json 2> $OUT/bench_gpus2_rehearsal.err || echo "bench --gpus 2 (rehearsal) failed"
PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --lean --flavour 1" bash tests/tools/pmc_extend.sh 0 0 gpurun_out/$TAG/pmc_batched_flavour1 > $OUT/pmc_batched_flavour1.txt 2>&1 || echo "pmc batched flavour 1: a pass failed"
rm -rf $OUT/pmc_batched_flavour1/p*/
CENSUS_FLAVOURS=1 bash tests/tools/trip_census.sh $TAG > $OUT/trip_census.log 2>&1 || echo "trip census failed"
python3 bench.py --steps 20 --warmup 5 --flavour 1 --seed-mode 1 --no-cpu-baseline --lean > $OUT/bench_reference_semantics_stale_model.json 2> $OUT/bench_reference_semantics.err || echo "bench flavour 1 / seed mode 1 failed"
ls $OUT
