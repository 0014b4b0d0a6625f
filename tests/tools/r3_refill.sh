#!/bin/bash
# Developer helper for gpurun (round 3): refill threshold (idle lanes: 24 / 32 / 40 / 48 / 56 = hundreds 7 / 9 / 10 / 11 / 12, 7 workgroups per CU = tens 5)
# on the scenes beyond the caches; interleaved rounds in one process per scene (tests/tools/ab_bench.py)
OUT=gpurun_out/${1:-r3r}
mkdir -p $OUT
for T in ${SCENES:-1000000 6000000}; do
  VARIANTS=${VARIANTS:-751,951,1051,1151,1251} MODE=batched ROUNDS=3 STEPS=4 SCENE=soup:$T timeout -k 10 900 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/soup:$T  /"
done | tee $OUT/refill.txt
