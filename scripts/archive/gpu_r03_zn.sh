#!/bin/bash
# round 3: table lookups of a sample's address one step ahead of its gather (-DVR_LUT_AHEAD=1; depth 5 and depth 4) against the product
set -e
mkdir -p gpurun_out/r03zn
VR_HIP_LIB=$PWD/build_variants/libvr_hip_ahead5.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -x -q -m gpu > gpurun_out/r03zn/pytest_ahead5.log 2>&1 || { tail -30 gpurun_out/r03zn/pytest_ahead5.log; exit 1; }
tail -2 gpurun_out/r03zn/pytest_ahead5.log
for v in product ahead5 ahead4 product; do
  if [ $v = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$v.so; fi
  python scripts/perf_probe.py --reps 6 >> gpurun_out/r03zn/$v.jsonl
  python scripts/perf_probe.py --reps 6 --mode default >> gpurun_out/r03zn/${v}_default.jsonl
done
