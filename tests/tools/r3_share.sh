#!/bin/bash
# Developer helper for gpurun (round 3): a rank's share of the 8-GPU step (259 200 photons x 8 waves in one fused launch) under
# different grid sizes (uvrt_set_variant 6g1: g = 0 8, 1 4, 2 6, 3 2, 5 7 workgroups per CU), and the same for 2- and 4-GPU shares
OUT=gpurun_out/${1:-r3o}
mkdir -p $OUT
for ph in 259200 518400 1036800; do
  PHOTONS=$ph VARIANTS=0,611,621,631,601 MODE=batched ROUNDS=4 STEPS=40 timeout -k 10 300 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/photons=$ph  /"
done | tee $OUT/share_grid.txt
