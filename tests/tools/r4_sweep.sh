#!/bin/bash
# round 4 (after the drain merge): refill threshold and grid size once more -- uvrt_set_variant 401 / 801 / 701 = refill at 16 / 4 / 24
# idle lanes, 621 / 651 / 611 = 6 / 7 / 4 workgroups per CU (0 = the default: 8 idle lanes, 8 or 7 per CU)
TAG=${1:-r4sweep}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for mode in batched loop loop_sync; do
  FLAVOURS=0,2 VARIANTS=0,401,801,701,621,651,611 MODE=$mode ROUNDS=4 STEPS=30 timeout -k 10 500 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant"
done | tee $OUT/sweep.txt
