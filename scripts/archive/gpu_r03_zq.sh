#!/bin/bash
# round 3: wave priority raised (s_setprio 1 / 3) while a sample's address chain and gather are issued, back to 0 for the rest of the sample
# (build_variants/libvr_hip_prio{1,3}.so from a temporary edit of the TRILINEAR loop; images unchanged)
set -e
mkdir -p gpurun_out/r03zq
for v in product prio1 prio3 product; do
  if [ $v = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$v.so; fi
  python scripts/perf_probe.py --reps 6 >> gpurun_out/r03zq/$v.jsonl
  python scripts/perf_probe.py --reps 6 --mode default >> gpurun_out/r03zq/${v}_default.jsonl
done
