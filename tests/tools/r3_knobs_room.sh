#!/bin/bash
# Developer probe (gpurun): batch chunk size and deposit replicas on the room, batched mode, alternating processes (two passes)
OUT=gpurun_out/${1:-r3u}
mkdir -p $OUT
for pass in 1 2; do
  for chunk in 64 96 128; do
    for repl in 8 12 16; do
      UVRT_BATCH_CHUNK_MB=$chunk UVRT_REPLICAS=$repl VARIANTS=0 MODE=batched ROUNDS=3 STEPS=30 timeout -k 10 200 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant" | sed "s/^/chunk=$chunk replicas=$repl  /"
    done
  done
done | tee $OUT/knobs_room.txt
