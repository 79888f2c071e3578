#!/bin/bash
# round 3: every view with the run bricks along z / along y forced, against the automatic copy choice
set -e
mkdir -p gpurun_out/r03zd
python scripts/perf_probe.py --reps 6 --plane 3 > gpurun_out/r03zd/runz.json
python scripts/perf_probe.py --reps 6 --plane 4 > gpurun_out/r03zd/runy.json
python scripts/perf_probe.py --reps 6 > gpurun_out/r03zd/auto.json
