#!/bin/bash
# tests + default-mode timings + 1-rank distributed bench
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/probe6
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/probe6/tests.log 2>&1 || { tail -30 gpurun_out/probe6/tests.log; exit 1; }
tail -2 gpurun_out/probe6/tests.log
python scripts/perf_probe.py --mode default
python scripts/perf_probe.py --mode default --sampling nearest
python scripts/perf_probe.py
bash scripts/gpu_bench_dist1.sh
