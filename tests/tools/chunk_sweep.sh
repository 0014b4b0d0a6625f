#!/bin/bash
# Developer probe: bench.py (batched mode) for several chunk sizes of uvrt_trace_batch (UVRT_BATCH_CHUNK_MB)
mkdir -p gpurun_out/sweep
for mb in ${CHUNKS:-34 70 100 140 300}; do
  UVRT_BATCH_CHUNK_MB=$mb python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline'] or {}
print('chunk MB $mb', d['value'], d['ms_per_step'], 'single', d['single_computation']['ms'], 'loop', d['other_modes']['loop']['value'], 'extend avg ms', r.get('avg_launch_ms'), r.get('rays_per_launch'))"
done 2>&1 | tee gpurun_out/sweep/chunks.txt
