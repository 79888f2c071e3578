#!/bin/bash
# round 3: three frames in flight (bench.py rank path rehearsed on one GPU, vr_hip_multi_*, scale_model)
set -e
mkdir -p gpurun_out/r03w
python -m pytest tests -x -q -m gpu -k "multi or distributed or launcher or scheduling" > gpurun_out/r03w/pytest.log 2>&1 || { tail -30 gpurun_out/r03w/pytest.log; exit 1; }
tail -3 gpurun_out/r03w/pytest.log
VR_BENCH_TWO_STREAMS=1 python bench.py --force-launcher --steps 48 --warmup 16 --no-extras --no-cpu-baseline > gpurun_out/r03w/launcher_three_slots.json
python bench.py --steps 24 --warmup 8 --no-cpu-baseline --extras scale,multi > gpurun_out/r03w/scale_multi.json
