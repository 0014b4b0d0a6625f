#!/bin/bash
# Developer helper for gpurun (round 3): ray ordering (uvrt_set_sort_bits, loop mode) and grid size on the scenes beyond the caches
OUT=gpurun_out/${1:-r3h}
mkdir -p $OUT
for T in ${SCENES:-1000000 6000000}; do
  for sb in 0 -1 12 16 20; do
    timeout -k 10 500 python3 bench.py --scene soup:$T --mode loop --sort-bits $sb --steps 3 --warmup 1 --no-cpu-baseline --lean 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline'] or {}
print('soup:$T loop sort_bits $sb', d['value'], 'ms/step', d['ms_per_step'], 'extend ms', r.get('avg_launch_ms'), d['dose_crc32'])"
  done
  for v in 0 601 621; do
    timeout -k 10 500 python3 bench.py --scene soup:$T --variant $v --steps 3 --warmup 1 --no-cpu-baseline --lean 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline'] or {}
print('soup:$T batched variant $v', d['value'], 'ms/step', d['ms_per_step'], 'extend ms', r.get('avg_launch_ms'), d['dose_crc32'])"
  done
done 2>&1 | tee $OUT/sort_grid.txt
