#!/bin/bash
# round 3: 2-byte voxels, perspective views: quad bricks (automatic) against oct bricks with the three wave shapes, and quad bricks with the wave shapes
set -e
mkdir -p gpurun_out/r03zx
python scripts/perf_probe.py --bpv 2 --reps 4 > gpurun_out/r03zx/auto.json
for lm in 2 6 10; do
  python scripts/perf_probe.py --bpv 2 --reps 4 --views 4,5,6,7 --plane 5 --tile-map $lm,0,0 > gpurun_out/r03zx/oct_$lm.json
  python scripts/perf_probe.py --bpv 2 --reps 4 --views 4,5,6,7 --tile-map $lm,0,0 > gpurun_out/r03zx/quad_$lm.json
done
