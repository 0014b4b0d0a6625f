#!/bin/bash
# round 4: the shipped-flags flavour (uvrt_set_flavour 2) against the default -- A/B in one process, the 24.9 M-ray parity run
# against the reference kernel built with its own flags, PMC passes of both flavours from the same build
TAG=${1:-r4f2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
FLAVOUR=2 timeout -k 10 500 python tests/tools/big_parity.py > $OUT/big_parity_flavour2.log 2>&1; echo "big parity rc=$?"; tail -4 $OUT/big_parity_flavour2.log
for mode in batched loop loop_sync; do
  FLAVOURS=0,2 VARIANTS=0 MODE=$mode ROUNDS=5 STEPS=20 ISOLATED=$([ $mode = loop ] && echo 1 || echo 0) timeout -k 10 300 python tests/tools/ab_bench.py 2>/dev/null | grep "^variant"
done | tee $OUT/ab_flavours.txt
if [ "${2:-}" != nopmc ]; then
  PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --lean" bash tests/tools/pmc_extend.sh 0 0 $OUT/pmc_flavour0 > $OUT/pmc_flavour0.txt 2>&1 || echo "pmc flavour 0: a pass failed"
  PMC_ARGS="--steps 3 --warmup 1 --no-cpu-baseline --lean --flavour 2" bash tests/tools/pmc_extend.sh 0 0 $OUT/pmc_flavour2 > $OUT/pmc_flavour2.txt 2>&1 || echo "pmc flavour 2: a pass failed"
  rm -rf $OUT/pmc_flavour0/p*/ $OUT/pmc_flavour2/p*/
  paste $OUT/pmc_flavour0/summary.txt $OUT/pmc_flavour2/summary.txt | awk '{printf "%-36s %16s %16s\n", $1, $4, $11}'
fi
