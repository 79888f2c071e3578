#!/bin/bash
# round 2, call K: fewer resident waves per CU (LDS padding) so that a wave's step-to-step line reuse fits the 32 KiB L1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02k; mkdir -p $OUT
BV=$GRAFT_REPO_ROOT/build_variants
for V in head pad24 pad16; do
  echo "== $V"; VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py --reps 6 || exit 1
  export VR_HIP_LIB=$BV/libvr_hip_$V.so
  bash scripts/gpu_pmc.sh $OUT/pmc_$V tcc,tcp1 --views 0,1,2,3,4,5,6,7 > /dev/null || exit 1
  python scripts/pmc_per_view.py $OUT/pmc_$V 3 | grep "RDREQ_sum\|TCC_HIT\|TCP_TCC_READ"
done
