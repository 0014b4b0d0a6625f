#!/bin/bash
# Developer probe: isolated extend time (one stream) of the library builds in tests/tools/_perturb/
mkdir -p gpurun_out/perturb
cp small-project-uv-robot-ray-tracer_amd/libuvrt_hip.so /tmp/libuvrt_keep.so
for tag in ${TAGS:-base}; do
  cp tests/tools/_perturb/libuvrt_hip_$tag.so small-project-uv-robot-ray-tracer_amd/libuvrt_hip.so
  echo "== $tag"
  VARIANTS=${VARIANTS:-0} SORTS=0 CHECK=0 ROUNDS=4 PIPELINE=0 timeout -k 10 200 python tests/tools/quick_extend_bench.py 2>&1 | grep "^variant"
done 2>&1 | tee gpurun_out/perturb/quick.txt
cp /tmp/libuvrt_keep.so small-project-uv-robot-ray-tracer_amd/libuvrt_hip.so
