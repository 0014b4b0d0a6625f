#!/bin/bash
# round 4: flavour 2 with 8 workgroups per CU in the pipelined modes (601: refill at 8 idle lanes, 401: at 16) against the default's 7 (651 / 0)
TAG=${1:-r4sweep3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for i in 1 2; do
for mode in batched loop loop_sync; do
  FLAVOURS=2 VARIANTS=651,601,401,0 MODE=$mode ROUNDS=4 STEPS=30 timeout -k 10 500 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant"
done
done | tee $OUT/sweep.txt
