#!/bin/bash
# round 3: lane order along the clean screen direction on orthogonal views along an axis — all views, all sampling modes, wave shapes on view 2
set -e
mkdir -p gpurun_out/r03zc
python scripts/perf_probe.py --reps 6 > gpurun_out/r03zc/tri.json
python scripts/perf_probe.py --reps 6 --sampling nearest > gpurun_out/r03zc/near.json
python scripts/perf_probe.py --reps 6 --mode default > gpurun_out/r03zc/tri_default.json
python scripts/perf_probe.py --reps 6 --views 2 --tile-map 0,7,0 > gpurun_out/r03zc/v2_s0.json
python scripts/perf_probe.py --reps 6 --views 2 --tile-map 4,7,0 > gpurun_out/r03zc/v2_s1.json
python scripts/perf_probe.py --reps 6 --views 2 --tile-map 8,7,0 > gpurun_out/r03zc/v2_s2.json
python scripts/perf_probe.py --reps 6 --views 2 --tile-map 8,7,4 > gpurun_out/r03zc/v2_s2b.json
python scripts/perf_probe.py --reps 6 --views 0 --tile-map 10,7,7 > gpurun_out/r03zc/v0_s2.json
python scripts/perf_probe.py --reps 6 --views 0 --tile-map 6,7,7 > gpurun_out/r03zc/v0_s1.json
python scripts/perf_probe.py --reps 6 --volume 512 --viewport 1024 > gpurun_out/r03zc/tri_512.json
