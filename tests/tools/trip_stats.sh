#!/bin/bash
# Developer probe (run on the GPU box; it rebuilds the library of the box's scratch copy with -DUVRT_TRIP_STATS):
# where the lanes of the traversal trips go -- inner / leaf / waiting / idle lanes per trip, drain-phase trips.
TAG=${1:-tripstats}
mkdir -p gpurun_out/$TAG
cd small-project-uv-robot-ray-tracer_amd && touch csrc/uvrt_extend6.hip && make -s libuvrt_hip.so EXTRA_HIPFLAGS=-DUVRT_TRIP_STATS=${LEVEL:-2} > ../gpurun_out/$TAG/build.log 2>&1 || { tail ../gpurun_out/$TAG/build.log; exit 1; }
cd ..
for v in ${VARIANTS:-0}; do
  echo "== variant $v, single launches" >> gpurun_out/$TAG/stats.log
  UVRT_TRIP_STATS=1 VARIANTS=$v SORTS=0 CHECK=0 ROUNDS=1 PIPELINE=0 timeout -k 10 200 python tests/tools/quick_extend_bench.py >> gpurun_out/$TAG/stats.log 2>&1
done
echo "== bench, batched" >> gpurun_out/$TAG/stats.log
UVRT_TRIP_STATS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --lean --steps 2 --warmup 0 >> gpurun_out/$TAG/stats.log 2>&1
grep -E "^==|trip stats|trip clocks|extend " gpurun_out/$TAG/stats.log | tail -30
