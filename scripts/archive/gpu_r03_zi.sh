#!/bin/bash
# round 3: workgroup tile stretched along the wave shape's long side (32x16 -> 16x32 -> 8x64 / 64x8 -> 128x4) on the run-brick views;
# needs scripts/ubench/tile_stretch_experiment.patch applied (the product rejects lane maps >= 16)
set -e
mkdir -p gpurun_out/r03zi
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_mapping or layouts_agree" > gpurun_out/r03zi/pytest.log 2>&1 || { tail -30 gpurun_out/r03zi/pytest.log; exit 1; }
tail -2 gpurun_out/r03zi/pytest.log
python scripts/perf_probe.py --reps 6 > gpurun_out/r03zi/auto.json
for lm in 10 26 42 6 22 38 2 18 34; do
  python scripts/perf_probe.py --reps 6 --views 4,6,7 --tile-map $lm,0,0 > gpurun_out/r03zi/persp_$lm.json
done
for lm in 1 17 33 9 25 41 5 21 37 0 16 32; do
  python scripts/perf_probe.py --reps 6 --views 1,3,5 --tile-map $lm,0,0 > gpurun_out/r03zi/obl_$lm.json
done
