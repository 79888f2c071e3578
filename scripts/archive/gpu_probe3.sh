#!/bin/bash
# tests + per-view timings (bricked layout)
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/probe3
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/probe3/tests.log 2>&1 || { tail -30 gpurun_out/probe3/tests.log; exit 1; }
tail -2 gpurun_out/probe3/tests.log
python scripts/perf_probe.py > gpurun_out/probe3/tri.json && cat gpurun_out/probe3/tri.json
python scripts/perf_probe.py --light 0 > gpurun_out/probe3/tri_nolight.json && cat gpurun_out/probe3/tri_nolight.json
python scripts/perf_probe.py --mode default > gpurun_out/probe3/tri_default.json && cat gpurun_out/probe3/tri_default.json
python scripts/perf_probe.py --volume 256 --viewport 1024 > gpurun_out/probe3/c2.json && cat gpurun_out/probe3/c2.json
