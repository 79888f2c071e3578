#!/bin/bash
# round 3: exhaustive (lane order x wave shape x phase) sweep per view against the automatic tile mapping
set -e
mkdir -p gpurun_out/r03za
python scripts/tile_map_sweep.py > gpurun_out/r03za/sweep_trilinear.jsonl
python scripts/tile_map_sweep.py --sampling nearest --lane-maps 0,1,2 > gpurun_out/r03za/sweep_nearest.jsonl
