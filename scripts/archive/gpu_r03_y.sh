#!/bin/bash
# round 3: three frames in flight with 8 hardware queues (bench.py sets GPU_MAX_HW_QUEUES before HIP starts)
set -e
mkdir -p gpurun_out/r03y
VR_BENCH_TWO_STREAMS=1 python bench.py --force-launcher --steps 48 --warmup 16 --no-extras --no-cpu-baseline > gpurun_out/r03y/launcher_three_slots.json
python bench.py --steps 24 --warmup 8 --no-cpu-baseline --extras scale,multi > gpurun_out/r03y/scale_multi.json
python bench.py --steps 24 --warmup 8 --no-cpu-baseline --extras modes > gpurun_out/r03y/modes.json
