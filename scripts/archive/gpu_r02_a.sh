#!/bin/bash
# round 2, call A: tests, kernel-variant A/B (build_variants/*.so via VR_HIP_LIB), per-view PMC of all 8 benchmark views at HEAD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02a; mkdir -p $OUT
BV=$GRAFT_REPO_ROOT/build_variants
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
VR_HIP_LIB=$BV/libvr_hip_asm.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $OUT/tests_asm.log 2>&1 || { tail -30 $OUT/tests_asm.log; echo "asm variant FAILED parity"; }
tail -1 $OUT/tests_asm.log
for V in base asm center rsq2; do
  echo "== $V lit";   VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py || exit 1
done
for V in base asm; do
  echo "== $V unlit"; VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py --light 0 || exit 1
done
echo "== base nearest lit"; VR_HIP_LIB=$BV/libvr_hip_base.so timeout -k 10 300 python scripts/perf_probe.py --sampling nearest || exit 1
echo "== base lit lane blocks forced"; VR_HIP_LIB=$BV/libvr_hip_base.so timeout -k 10 300 python scripts/perf_probe.py --views 1,5 --tile-map 2,0,0 || exit 1
export VR_HIP_LIB=$BV/libvr_hip_base.so
bash scripts/gpu_pmc.sh $OUT/pmc sq1,sq2,tcp1,tcc,fetch --views 0,1,2,3,4,5,6,7 || exit 1
python scripts/pmc_per_view.py $OUT/pmc 3 | tee $OUT/pmc_per_view.txt
