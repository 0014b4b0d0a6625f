#!/bin/bash
# Developer probe: bench.py for several deposit-replica counts (UVRT_REPLICAS)
mkdir -p gpurun_out/sweep
for r in ${REPLICAS:-64 32 16 8 64}; do
  UVRT_REPLICAS=$r python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('replicas $r batched', d['value'], 'loop', d['other_modes']['loop']['value'], 'single', d['single_computation']['mray_s'], d['dose_crc32'])"
done 2>&1 | tee gpurun_out/sweep/replicas.txt
