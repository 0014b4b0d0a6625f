#!/bin/bash
# HBM traffic of the extend kernel for bench.py's roofline.traffic (MI355X_MICROARCH.md "HBM"):
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (3 + 2 TCC slots), kilobytes per dispatch;
# on gfx950 FETCH_SIZE counts 128-byte fabric requests as 64 bytes, so the read side is doubled.
# Writes profiles/extend_pmc.json.   Usage on the GPU box: bash tests/tools/pmc_traffic.sh
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  tag=$(echo $C | cut -d' ' -f1)
  VARIANTS=0 SORTS=0 CHECK=0 timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/$tag -- python3 $REPO/tests/tools/quick_extend_bench.py > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_extend" in row["Kernel_Name"]:
            a = agg[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
avg = {k: v[0] / v[1] for k, v in agg.items()}
fetch_kb, write_kb = avg.get("FETCH_SIZE", 0.0), avg.get("WRITE_SIZE", 0.0)
out = {"kernel": "k_extend6<2, false, true> (default)", "rays_per_launch": 2073600, "launches_averaged": int(agg["FETCH_SIZE"][1]),
       "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "raw": avg,
       "method": "rocprofv3 --pmc, one pass per counter; read bytes = FETCH_SIZE*1024*2 (gfx950: 128-B requests "
                 "tallied as 64 B), write bytes = WRITE_SIZE*1024 (each deposit atomic counts as one 32-B write)",
       "hbm_bytes_per_launch": fetch_kb * 1024 * 2 + write_kb * 1024}
json.dump(out, open("$REPO/profiles/extend_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
