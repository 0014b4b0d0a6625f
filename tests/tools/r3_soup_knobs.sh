#!/bin/bash
# Developer helper for gpurun (round 3): kernel knobs on the scenes beyond the caches, developer build swapped in
# (leaf period 1 / 2 / 3 = last digit 0 / 1 / 2; refill threshold 16 / 8 / 24 / 4 = hundreds 4 / 6 / 7 / 8; 7 workgroups per CU = tens 5)
OUT=gpurun_out/${1:-r3p}
mkdir -p $OUT
P=small-project-uv-robot-ray-tracer_amd
cp $P/libuvrt_hip.so /tmp/libuvrt_keep.so
cp $P/libuvrt_hip_dev.so $P/libuvrt_hip.so
for T in ${SCENES:-1000000 6000000}; do
  VARIANTS=0,650,652,451,751,851,850 MODE=batched ROUNDS=3 STEPS=4 SCENE=soup:$T timeout -k 10 600 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/soup:$T  /"
done | tee $OUT/soup_knobs.txt
cp /tmp/libuvrt_keep.so $P/libuvrt_hip.so
