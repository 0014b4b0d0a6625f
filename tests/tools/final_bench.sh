#!/bin/bash
# Developer helper for gpurun: the three bench lines of the round (after profiles/extend_issue_model_*.json were refreshed)
OUT=gpurun_out/final
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_batched.json 2> $OUT/bench_batched.err || echo "bench batched failed"
python3 bench.py --steps 20 --warmup 5 --mode loop > $OUT/bench_loop.json 2> $OUT/bench_loop.err || echo "bench loop failed"
python3 bench.py --steps 20 --warmup 5 --flavour 1 > $OUT/bench_flavour1.json 2> $OUT/bench_flavour1.err || echo "bench flavour 1 failed"
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "bench default failed"
tail -c 300 $OUT/bench_default.json
