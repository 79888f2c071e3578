#!/bin/bash
# round 2, call I: pipeline depth sweep with managed loads (parity on the in-tree build, timing per depth)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02i; mkdir -p $OUT
BV=$GRAFT_REPO_ROOT/build_variants
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -5 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
for seed in 21 22; do VR_TEST_SEED=$seed timeout -k 10 600 python -m pytest tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -1; done
for D in 2 3 4 5; do
  echo "== depth $D trilinear"; VR_HIP_LIB=$BV/libvr_hip_depth$D.so timeout -k 10 300 python scripts/perf_probe.py || exit 1
  echo "== depth $D nearest";   VR_HIP_LIB=$BV/libvr_hip_depth$D.so timeout -k 10 300 python scripts/perf_probe.py --sampling nearest || exit 1
done
