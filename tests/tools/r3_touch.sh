#!/bin/bash
# Developer helper for gpurun (round 3): the touch of pushed records (variant 901 on / 900 off), parity first, then interleaved
# A/B on the room and on the soups, then the memory counters of soup:6M with the touch on.
OUT=gpurun_out/${1:-r3g}
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_adversarial.py tests/test_gpu_stress.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
VARIANTS=900,901 MODE=batched ROUNDS=4 STEPS=20 timeout -k 10 300 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/room        /" | tee $OUT/ab_touch.txt
for T in 300000 1000000 6000000; do
  VARIANTS=900,901 MODE=batched ROUNDS=3 STEPS=4 SCENE=soup:$T timeout -k 10 500 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/soup:$T  /" | tee -a $OUT/ab_touch.txt
done
