#!/bin/bash
# round 4: confirmation of the final tree on a fresh box: whole GPU suite, smoke, fuzz soak, the 24.9 M-ray parity runs, two lines
TAG=${1:-r4confirm}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "suite rc=$?"; tail -3 $OUT/tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
UVRT_FUZZ_CASES=400 timeout -k 10 600 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > $OUT/fuzz_soak_400.txt 2>&1; echo "fuzz rc=$?"; tail -2 $OUT/fuzz_soak_400.txt
timeout -k 10 400 python tests/tools/big_parity.py > $OUT/big_parity_flavour0.log 2>&1; echo "big parity fl0 rc=$?"; tail -3 $OUT/big_parity_flavour0.log
FLAVOUR=2 timeout -k 10 400 python tests/tools/big_parity.py > $OUT/big_parity_flavour2.log 2>&1; echo "big parity fl2 rc=$?"; tail -3 $OUT/big_parity_flavour2.log
python3 bench.py --steps 20 --warmup 5 --flavour 1 --seed-mode 1 --no-cpu-baseline --lean > $OUT/bench_reference_semantics.json 2> $OUT/bench_reference_semantics.err || echo "bench flavour 1 / seed mode 1 failed"
python3 bench.py > $OUT/bench_noflags.json 2> $OUT/bench_noflags.err || echo "bench (no flags) failed"
cut -c1-160 $OUT/bench_noflags.json
