#!/bin/bash
# round 4: whole GPU suite, then old / new library A/B in both flavours (alternating processes)
TAG=${1:-r4full}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1; echo "suite rc=$?"; tail -4 $OUT/tests.log
P=small-project-uv-robot-ray-tracer_amd
cp $P/libuvrt_hip.so /tmp/libuvrt_new.so
for i in 1 2 3; do
  for tag in new old; do
    if [ $tag = new ]; then cp /tmp/libuvrt_new.so $P/libuvrt_hip.so; else cp tests/tools/_ab/libuvrt_hip_old.so $P/libuvrt_hip.so; fi
    for mode in batched loop loop_sync; do
      FLAVOURS=0,2 VARIANTS=0 MODE=$mode ROUNDS=3 STEPS=30 timeout -k 10 300 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/$tag  /"
    done
  done
done | sort | tee $OUT/ab.txt
cp /tmp/libuvrt_new.so $P/libuvrt_hip.so
