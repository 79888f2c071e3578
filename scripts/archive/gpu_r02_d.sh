#!/bin/bash
# round 2, call D: 8-phase / exact-arithmetic tile phase (host only): parity, timing, traffic
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02d; mkdir -p $OUT
BV=$GRAFT_REPO_ROOT/build_variants
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
for V in sat phase8; do
  echo "== $V lit"; VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py || exit 1
done
echo "== phase8 nearest"; VR_HIP_LIB=$BV/libvr_hip_phase8.so timeout -k 10 300 python scripts/perf_probe.py --sampling nearest || exit 1
export VR_HIP_LIB=$BV/libvr_hip_phase8.so
bash scripts/gpu_pmc.sh $OUT/pmc tcc,fetch,tcp1,sq2 --views 0,1,2,3,4,5,6,7 || exit 1
python scripts/pmc_per_view.py $OUT/pmc 3 | tee $OUT/pmc_per_view.txt
