#!/bin/bash
# round 2, call C: brick order with the march axis slowest (parity + timing); XCD mappings re-measured with traffic counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02c; mkdir -p $OUT
BV=$GRAFT_REPO_ROOT/build_variants
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
for V in sat order xcd1 xcd2; do
  echo "== $V lit"; VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py || exit 1
done
for V in order xcd1 xcd2; do
  export VR_HIP_LIB=$BV/libvr_hip_$V.so
  bash scripts/gpu_pmc.sh $OUT/pmc_$V tcc,fetch --views 0,1,2,3,4,5,6,7 || exit 1
  echo "-- $V"; python scripts/pmc_per_view.py $OUT/pmc_$V 3 | tee $OUT/pmc_$V.txt
done
