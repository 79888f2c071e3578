#!/bin/bash
# round 3: automatic tile mapping against the whole 8 x 8 phase grid on the orthogonal views along an axis
set -e
mkdir -p gpurun_out/r03zb
python scripts/phase_grid_probe.py --views 0,2,3 --lane-maps 2,0 > gpurun_out/r03zb/grid.jsonl
