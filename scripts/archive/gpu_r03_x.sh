#!/bin/bash
# round 3: do concurrent slot streams depend on how many OTHER streams of the process hold a hardware queue (GPU_MAX_HW_QUEUES, default 4)?
set -e
mkdir -p gpurun_out/r03x
python scripts/overlap_probe.py --ranks 8 --streams 1,2,3,4 --skip-streams 0 > gpurun_out/r03x/skip0.json
python scripts/overlap_probe.py --ranks 8 --streams 1,2,3,4 --skip-streams 2 > gpurun_out/r03x/skip2.json
python scripts/overlap_probe.py --ranks 8 --streams 1,2,3,4 --skip-streams 4 > gpurun_out/r03x/skip4.json
GPU_MAX_HW_QUEUES=8 python scripts/overlap_probe.py --ranks 8 --streams 1,2,3,4,6 --skip-streams 0 > gpurun_out/r03x/q8_skip0.json
GPU_MAX_HW_QUEUES=8 python scripts/overlap_probe.py --ranks 8 --streams 1,2,3,4,6 --skip-streams 2 > gpurun_out/r03x/q8_skip2.json
GPU_MAX_HW_QUEUES=8 python scripts/overlap_probe.py --ranks 2,4 --streams 1,2,3,4 --skip-streams 2 > gpurun_out/r03x/q8_skip2_n24.json
