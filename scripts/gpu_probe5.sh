#!/bin/bash
# tests + NEAREST timings
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/probe5
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/probe5/tests.log 2>&1 || { tail -30 gpurun_out/probe5/tests.log; exit 1; }
tail -2 gpurun_out/probe5/tests.log
python scripts/perf_probe.py --sampling nearest
python scripts/perf_probe.py --sampling nearest --light 0
python scripts/perf_probe.py --sampling nearest --mode default
