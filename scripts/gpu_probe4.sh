#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/probe4
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/probe4/tests.log 2>&1 || { tail -30 gpurun_out/probe4/tests.log; exit 1; }
tail -2 gpurun_out/probe4/tests.log
python scripts/perf_probe.py > gpurun_out/probe4/tri.json && cat gpurun_out/probe4/tri.json
python scripts/perf_probe.py --sampling nearest > gpurun_out/probe4/near.json && cat gpurun_out/probe4/near.json
python scripts/perf_probe.py --mode default > gpurun_out/probe4/tri_default.json && cat gpurun_out/probe4/tri_default.json
python scripts/perf_probe.py --mode default --sampling nearest > gpurun_out/probe4/near_default.json && cat gpurun_out/probe4/near_default.json
python scripts/perf_probe.py --mode default --sampling nearest --layout linear > gpurun_out/probe4/near_default_lin.json && cat gpurun_out/probe4/near_default_lin.json
