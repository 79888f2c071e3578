#!/bin/bash
# round 3: per-tile cost maps of the oblique views under the two run-brick copies (does the better copy depend on the cube face a tile's rays enter through?)
set -e
mkdir -p gpurun_out/r03z
python scripts/cost_map_probe.py --view 1 --planes 3,4 --save gpurun_out/r03z/view1.npz > gpurun_out/r03z/view1.json
python scripts/cost_map_probe.py --view 5 --planes 3,4 --save gpurun_out/r03z/view5.npz > gpurun_out/r03z/view5.json
python scripts/cost_map_probe.py --view 3 --planes 3,4,1 > gpurun_out/r03z/view3.json
