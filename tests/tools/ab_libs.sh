#!/bin/bash
# Developer probe (gpurun): A/B of two builds of libuvrt_hip.so in alternating processes on one box
#   bash tests/tools/ab_libs.sh <outdir> <other.so> [rounds]     (the in-tree product library is "new", <other.so> is "old")
OUT=gpurun_out/${1:-ab}
OTHER=${2:-tests/tools/_ab/libuvrt_hip_old.so}
R=${3:-3}
mkdir -p $OUT
P=small-project-uv-robot-ray-tracer_amd
cp $P/libuvrt_hip.so /tmp/libuvrt_new.so
for i in $(seq 1 $R); do
  for tag in new old; do
    if [ $tag = new ]; then cp /tmp/libuvrt_new.so $P/libuvrt_hip.so; else cp $OTHER $P/libuvrt_hip.so; fi
    for mode in ${MODES:-batched loop}; do
      VARIANTS=0 MODE=$mode ROUNDS=3 STEPS=30 ISOLATED=$([ $mode = loop ] && echo 1 || echo 0) timeout -k 10 200 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/$tag  /"
    done
  done
done | tee $OUT/ab_libs.txt
cp /tmp/libuvrt_new.so $P/libuvrt_hip.so
