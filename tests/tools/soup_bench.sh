#!/bin/bash
# Developer helper for gpurun: bench lines for synthetic scenes beyond L2 / the Infinity Cache (bench.py --scene soup:T)
mkdir -p gpurun_out/soup
for spec in "300000 5 2" "1000000 5 2" "6000000 3 1"; do
  set -- $spec
  timeout -k 10 500 python3 bench.py --scene soup:$1 --mode loop --steps $2 --warmup $3 > gpurun_out/soup/soup$1.json 2> gpurun_out/soup/soup$1.err || { echo "soup $1 failed"; tail -3 gpurun_out/soup/soup$1.err; }
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/soup/soup$1.json").read().strip().splitlines()[-1])
    print("soup $1", d["value"], d.get("other_modes"), d["dose_crc32"], (d.get("cpu_baseline") or {}).get("gpu_dose_bit_identical"))
except Exception as e:
    print("soup $1: no line", e)
PY
done
