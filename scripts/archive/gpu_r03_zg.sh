#!/bin/bash
# round 3: how much cheaper must the copy along y be for a block of tiles to switch (kLayoutRunDual)?
set -e
mkdir -p gpurun_out/r03zg
for k in 0 80 90 95 97; do
  VR_DUAL_KEEP_PERCENT=$k python scripts/perf_probe.py --reps 8 --views 1,5 > gpurun_out/r03zg/keep$k.json
done
python scripts/perf_probe.py --reps 8 --views 1,5 --plane 3 > gpurun_out/r03zg/runz.json
