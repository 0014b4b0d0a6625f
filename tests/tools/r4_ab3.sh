#!/bin/bash
# A/B/C: in-tree library ("new") against tests/tools/_ab/libuvrt_hip_{3site,old}.so, alternating processes, two flavours
TAG=${1:-r4ab3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
P=small-project-uv-robot-ray-tracer_amd
cp $P/libuvrt_hip.so /tmp/libuvrt_new.so
for i in 1 2 3; do
  for tag in new ${OTHERS:-3site old}; do
    if [ $tag = new ]; then cp /tmp/libuvrt_new.so $P/libuvrt_hip.so; else cp tests/tools/_ab/libuvrt_hip_$tag.so $P/libuvrt_hip.so; fi
    for mode in ${MODES:-batched loop loop_sync}; do
      FLAVOURS=0,2 VARIANTS=0 MODE=$mode ROUNDS=3 STEPS=30 timeout -k 10 300 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/$tag  /"
    done
  done
done | sort | tee $OUT/ab.txt
cp /tmp/libuvrt_new.so $P/libuvrt_hip.so
