#!/bin/bash
# round 4: the drain merge at 8 / 12 / 16 rays per wave (libraries built with -DUVRT_MERGE6_AT=k), alternating processes
TAG=${1:-r4mergeat}
OUT=gpurun_out/$TAG
mkdir -p $OUT
P=small-project-uv-robot-ray-tracer_amd
cp $P/libuvrt_hip.so /tmp/libuvrt_new.so
for i in 1 2; do
  for k in 16 12 8; do
    if [ $k = 16 ]; then cp /tmp/libuvrt_new.so $P/libuvrt_hip.so; else cp tests/tools/_ab/libuvrt_hip_m$k.so $P/libuvrt_hip.so; fi
    for mode in batched loop loop_sync; do
      FLAVOURS=0,2 VARIANTS=0 MODE=$mode ROUNDS=3 STEPS=30 timeout -k 10 300 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/at=$k  /"
    done
  done
done | sort | tee $OUT/ab.txt
cp /tmp/libuvrt_new.so $P/libuvrt_hip.so
