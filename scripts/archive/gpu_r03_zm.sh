#!/bin/bash
# round 3: resident waves per CU on the run-brick views (unused dynamic LDS: 0 -> 4 workgroups = 32 waves, 16 KiB (product) -> 3 = 24, 28 / 40 KiB -> 2 = 16)
set -e
mkdir -p gpurun_out/r03zm
for v in product pad0 pad28k pad40k; do
  if [ $v = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$v.so; fi
  python scripts/perf_probe.py --reps 6 --views 1,3,4,5,6,7 > gpurun_out/r03zm/$v.json
done
