#!/bin/bash
# round 4, first GPU call: the new tests (shipped flags, stale records, self-launching bench), then the whole suite, then the bench line
TAG=${1:-r4a}
mkdir -p gpurun_out/$TAG
timeout -k 10 500 python -m pytest tests/test_gpu_shipped_flags.py -m gpu -x -q -s > gpurun_out/$TAG/shipped.log 2>&1; echo "shipped rc=$?"; tail -25 gpurun_out/$TAG/shipped.log
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py tests/test_gpu_bench_ranks.py -m gpu -x -q > gpurun_out/$TAG/batch_ranks.log 2>&1; echo "batch/ranks rc=$?"; tail -15 gpurun_out/$TAG/batch_ranks.log
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/$TAG/tests.log 2>&1; echo "suite rc=$?"; tail -8 gpurun_out/$TAG/tests.log
timeout -k 10 400 python bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/$TAG/bench.err
python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/$TAG/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms", d["ms_per_step"])
    for k, v in d["other_modes"].items(): print(" ", k, v.get("value"), v.get("ms_per_step"), v.get("dose_crc32"), v.get("dose_crc32_expected"), v.get("triangles_beyond_1e-4_of_flavour0"))
    print("ref on gpu", json.dumps(d["cpu_baseline"]["reference_extend_cl_on_this_gpu"]))
except Exception as e:
    print("no bench line:", e)
PY
