#!/bin/bash
# round 4: A/B of the in-tree library against tests/tools/_ab/libuvrt_hip_old.so after a subset of the GPU tests
TAG=${1:-r4ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_golden.py tests/test_gpu_batch.py tests/test_gpu_fuzz.py tests/test_gpu_stress.py tests/test_gpu_pipeline.py tests/test_gpu_shipped_flags.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -4 $OUT/tests.log
MODES="${MODES:-batched loop loop_sync}" bash tests/tools/ab_libs.sh $TAG tests/tools/_ab/libuvrt_hip_old.so ${ROUNDS:-2} | sort
