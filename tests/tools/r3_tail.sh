#!/bin/bash
# Developer helper for gpurun (round 3): straggler rule of k_visit_stats (UVRT_HOT_TAIL) -- quality of the selection, the set-up
# kernels alone (one lamp per launch, 12 lamps), and the bench value
OUT=gpurun_out/${1:-r3t}
mkdir -p $OUT
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
for thr in ${THREADS:-256}; do for tl in ${TAILS:-6 12 16 24 32}; do
  export UVRT_HOT_TAIL=$tl UVRT_HOT_THREADS=$thr
  tail=${thr}_$tl
  echo "== UVRT_HOT_THREADS=$thr UVRT_HOT_TAIL=$tl"
  timeout -k 10 300 python -m pytest tests/test_gpu_hotset.py -m gpu -q > $OUT/tests_$tail.log 2>&1; tail -1 $OUT/tests_$tail.log
  timeout -k 10 300 python3 tests/tools/hotset_probe.py 2>&1 | tail -1
  (cd /tmp && export TMPDIR=/tmp && QUALITY=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/trace_$tail -- python3 $REPO/tests/tools/hotset_probe.py > $REPO/$OUT/trace_$tail.log 2>&1) || echo "trace failed"
  s=$(ls $OUT/trace_$tail/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$s" ] && cp $s $OUT/kernel_stats_tail$tail.csv && grep -E "visit_stats|select_hot|write_perm" $OUT/kernel_stats_tail$tail.csv | sed -e 's/(.*)"/"/' 
  rm -rf $OUT/trace_$tail
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('bench value', d['value'], d['dose_crc32'])"
done; done 2>&1 | tee $OUT/tail.txt
