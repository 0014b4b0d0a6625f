#!/bin/bash
# Developer probe: whole-step sensitivity to extra instructions per traversal trip.  tests/tools/_perturb/ holds
# builds of the library with N dummy vector / scalar instructions at the top of the common step.
mkdir -p gpurun_out/perturb
cp small-project-uv-robot-ray-tracer_amd/libuvrt_hip.so /tmp/libuvrt_keep.so
for tag in ${TAGS:-base v16 v32 s16 s32 base}; do
  cp tests/tools/_perturb/libuvrt_hip_$tag.so small-project-uv-robot-ray-tracer_amd/libuvrt_hip.so
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$tag batched', d['value'], 'loop', d['other_modes']['loop']['value'], 'single', d['single_computation']['mray_s'], d['dose_crc32'])"
done 2>&1 | tee gpurun_out/perturb/result.txt
cp /tmp/libuvrt_keep.so small-project-uv-robot-ray-tracer_amd/libuvrt_hip.so
