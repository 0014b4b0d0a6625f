#!/bin/bash
# Developer helper for gpurun: kernel parity tests, then the bench line and the isolated extend time.
TAG=${1:-quick}
mkdir -p gpurun_out/$TAG
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_adversarial.py tests/test_gpu_stress.py tests/test_gpu_batch.py -m gpu -x -q > gpurun_out/$TAG/tests.log 2>&1 || { tail -20 gpurun_out/$TAG/tests.log; exit 1; }
tail -2 gpurun_out/$TAG/tests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { tail -5 gpurun_out/$TAG/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
print("batched", d["value"], "other", d.get("other_modes"), "single", d.get("single_computation"), "crc", d.get("dose_crc32") or d.get("parity"))
print("roofline", d["roofline"].get("avg_launch_ms"), d["roofline"].get("frac"))
PY
SORTS=0 CHECK=1 ROUNDS=3 timeout -k 10 200 python tests/tools/quick_extend_bench.py > gpurun_out/$TAG/quick.log 2>&1; tail -4 gpurun_out/$TAG/quick.log
