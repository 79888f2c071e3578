#!/bin/bash
# round 2, call L: order of the cell columns inside a run brick (2-D Morton / x fastest / y fastest): parity, time, fabric requests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02l; mkdir -p $OUT
BV=$GRAFT_REPO_ROOT/build_variants
for V in runx runy; do VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "layouts_agree or trilinear_bit_exact" 2>&1 | tail -1; done
for V in head runx runy; do
  echo "== $V"; VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py --reps 6 || exit 1
  export VR_HIP_LIB=$BV/libvr_hip_$V.so
  bash scripts/gpu_pmc.sh $OUT/pmc_$V tcc --views 0,1,2,3,4,5,6,7 > /dev/null || exit 1
  python scripts/pmc_per_view.py $OUT/pmc_$V 3 | grep "RDREQ_sum"
  echo "-- $V run bricks forced"; VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py --reps 4 --plane 3 || exit 1
done
