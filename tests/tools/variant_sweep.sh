#!/bin/bash
# Developer probe: bench.py (loop mode, no CPU leg) for a list of extend kernel knob settings (uvrt_set_variant)
mkdir -p gpurun_out/sweep
for v in ${VARIANTS:-0 601 801 701 401 411 421 451 621 641}; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --variant $v 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline'] or {}
print('variant $v', d['value'], d['ms_per_step'], 'single', d['single_computation']['mray_s'], 'extend ms', r.get('avg_launch_ms'), d['dose_crc32'])"
done 2>&1 | tee gpurun_out/sweep/variants.txt
