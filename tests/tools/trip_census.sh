#!/bin/bash
# Developer probe (GPU box): trip counts by kind for tests/tools/stream_census.py.  The product library of the box's scratch copy
# is rebuilt with -DUVRT_TRIP_STATS (the C++ form of the loop with counters; same trips as the product's stream) and restored.
TAG=${1:-census}
OUT=gpurun_out/$TAG
mkdir -p $OUT
P=small-project-uv-robot-ray-tracer_amd
cp $P/libuvrt_hip.so /tmp/libuvrt_product.so
(cd $P && touch csrc/uvrt_extend6.hip csrc/uvrt_capi.hip && make -s libuvrt_hip.so EXTRA_HIPFLAGS=-DUVRT_TRIP_STATS=2 > ../$OUT/build.log 2>&1) || { tail $OUT/build.log; exit 1; }
for fl in ${CENSUS_FLAVOURS:-0 2}; do
  for mode in batched loop; do
    UVRT_TRIP_STATS=1 FLAVOUR=$fl MODE=$mode PIPELINE=0 COMPUTATIONS=2 timeout -k 10 200 python tests/tools/trip_census_run.py 2> $OUT/census_${mode}_f$fl.txt > /dev/null || echo "census $mode f$fl failed"
    grep -c "trip census" $OUT/census_${mode}_f$fl.txt
  done
done
cp /tmp/libuvrt_product.so $P/libuvrt_hip.so
(cd $P && touch csrc/uvrt_extend6.hip csrc/uvrt_capi.hip)
grep -h "trip census\|census rays" $OUT/census_batched_f0.txt | tail -3
