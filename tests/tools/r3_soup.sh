#!/bin/bash
# Developer helper for gpurun (round 3): scenes beyond the caches with the FINAL kernel -- bench lines in batched and loop mode,
# the opt-in 4-wide walk (loop mode only), and the memory counters of the batched extend launches (separate --pmc passes).
#   bash tests/tools/r3_soup.sh <outdir> <triangles> <steps> <warmup>
OUT=gpurun_out/${1:-r3d}
T=${2:-300000}; STEPS=${3:-5}; WARM=${4:-2}
mkdir -p $OUT
timeout -k 10 600 python3 bench.py --scene soup:$T --steps $STEPS --warmup $WARM > $OUT/soup${T}_batched.json 2> $OUT/soup${T}_batched.err || { echo "soup $T batched failed"; tail -3 $OUT/soup${T}_batched.err; }
timeout -k 10 600 python3 bench.py --scene soup:$T --mode loop --wide --steps $STEPS --warmup $WARM --no-cpu-baseline --lean > $OUT/soup${T}_wide_loop.json 2> $OUT/soup${T}_wide_loop.err || { echo "soup $T wide failed"; tail -3 $OUT/soup${T}_wide_loop.err; }
python3 - <<PY
import json
for tag in ("batched", "wide_loop"):
    try:
        d=json.loads(open("$OUT/soup${T}_%s.json" % tag).read().strip().splitlines()[-1])
        r=d.get("roofline") or {}
        print("soup $T", tag, d["value"], {k: v["value"] for k, v in (d.get("other_modes") or {}).items()}, d["dose_crc32"], (d.get("cpu_baseline") or {}).get("gpu_dose_bit_identical"), "launch ms", r.get("avg_launch_ms"), "rays/launch", r.get("rays_per_launch"))
    except Exception as e:
        print("soup $T", tag, "no line", e)
PY
PMC_ARGS="--scene soup:$T --steps 2 --warmup 1 --no-cpu-baseline --lean" PMC_PASSES="FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum|TCP_TOTAL_CACHE|SQ_WAIT_ANY|SQ_WAVES" bash tests/tools/pmc_extend.sh 0 0 $OUT/pmc_soup$T > $OUT/pmc_soup$T.txt 2>&1 || echo "pmc soup $T: a pass failed"
rm -rf $OUT/pmc_soup$T/p*/
grep -E "FETCH|WRITE|TCC_HIT|TCC_MISS|TCC_REQ|TCP_T|WAVE_CYCLES|WAIT_ANY|SQ_WAVES" $OUT/pmc_soup$T/summary.txt 2>/dev/null
