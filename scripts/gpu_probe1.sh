#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/probe1
rocprofv3 -L > gpurun_out/probe1/counters.txt 2>&1 || true
python scripts/perf_probe.py > gpurun_out/probe1/views_tri.json 2>gpurun_out/probe1/err.txt && cat gpurun_out/probe1/views_tri.json
python scripts/perf_probe.py --light 0 > gpurun_out/probe1/views_tri_nolight.json 2>>gpurun_out/probe1/err.txt && cat gpurun_out/probe1/views_tri_nolight.json
python scripts/perf_probe.py --sampling nearest > gpurun_out/probe1/views_near.json 2>>gpurun_out/probe1/err.txt && cat gpurun_out/probe1/views_near.json
python scripts/perf_probe.py --mode default > gpurun_out/probe1/views_tri_default.json 2>>gpurun_out/probe1/err.txt && cat gpurun_out/probe1/views_tri_default.json
python scripts/perf_probe.py --volume 256 --viewport 1024 > gpurun_out/probe1/views_c2.json 2>>gpurun_out/probe1/err.txt && cat gpurun_out/probe1/views_c2.json
