#!/bin/bash
# round 2, call J: which tiles of an 8x8-tile block an XCD owns (column of 8 / 2x4 / 4x2 / row of 8): time + fabric requests per view
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02j; mkdir -p $OUT
BV=$GRAFT_REPO_ROOT/build_variants
VR_HIP_LIB=$BV/libvr_hip_patch2.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nearest_bit_exact or partition" 2>&1 | tail -1
for V in head patch2 patch4 patch8; do
  echo "== $V"; VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py --reps 6 || exit 1
  export VR_HIP_LIB=$BV/libvr_hip_$V.so
  bash scripts/gpu_pmc.sh $OUT/pmc_$V tcc --views 0,1,2,3,4,5,6,7 > /dev/null || exit 1
  python scripts/pmc_per_view.py $OUT/pmc_$V 3 | grep "RDREQ_sum\|TCC_HIT"
done
