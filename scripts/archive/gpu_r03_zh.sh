#!/bin/bash
# round 3: GPU tier + bench after the per-block run-copy choice
set -e
mkdir -p gpurun_out/r03zh
python -m pytest tests -x -q -m gpu > gpurun_out/r03zh/pytest.log 2>&1 || { tail -30 gpurun_out/r03zh/pytest.log; exit 1; }
tail -2 gpurun_out/r03zh/pytest.log
python bench.py > gpurun_out/r03zh/bench.json
