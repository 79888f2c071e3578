#!/bin/bash
# round 3: what binds the march — timing-only variants that add two slow-rate / two fast-rate vector instructions / one LDS read per sample
# (build_variants/libvr_hip_{dslow,dfast,dlds}.so, built with -DVR_EXP_DUMMY_* from a temporary edit of tri_issue; images unchanged)
set -e
mkdir -p gpurun_out/r03zk
for v in product dslow dfast dlds; do
  if [ $v = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$v.so; fi
  python scripts/perf_probe.py --reps 6 > gpurun_out/r03zk/$v.json
  python scripts/perf_probe.py --reps 6 --light 0 > gpurun_out/r03zk/${v}_unlit.json
done
