#!/bin/bash
# round 4: refill at 16 idle lanes (451: 7 workgroups per CU, 401: 8) against 8 (651 = the default's setting in the pipelined modes), the
# default itself last (the first variant of a process measures low)
TAG=${1:-r4sweep2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for i in 1 2; do
for mode in batched loop; do
  FLAVOURS=0,2 VARIANTS=651,451,401,0 MODE=$mode ROUNDS=4 STEPS=30 timeout -k 10 500 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant"
done
done | tee $OUT/sweep.txt
