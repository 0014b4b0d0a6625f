#!/bin/bash
# round 4: the drain merge forced off / on, new library, alternating processes
TAG=${1:-r4knob}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for i in 1 2; do
  for k in 0 1; do
    for mode in ${MODES:-loop_sync loop batched}; do
      UVRT_DRAIN_MERGE=$k FLAVOURS=0 VARIANTS=0 MODE=$mode ROUNDS=3 STEPS=30 timeout -k 10 300 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/merge=$k  /"
    done
  done
done | sort | tee $OUT/ab.txt
