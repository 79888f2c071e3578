#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/probe2
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/probe2/tests.log 2>&1 || { tail -30 gpurun_out/probe2/tests.log; exit 1; }
tail -2 gpurun_out/probe2/tests.log
for L in bricked linear; do
python scripts/perf_probe.py --layout $L > gpurun_out/probe2/tri_$L.json && cat gpurun_out/probe2/tri_$L.json
python scripts/perf_probe.py --layout $L --light 0 > gpurun_out/probe2/tri_nolight_$L.json && cat gpurun_out/probe2/tri_nolight_$L.json
done
python scripts/perf_probe.py --mode default > gpurun_out/probe2/tri_default.json && cat gpurun_out/probe2/tri_default.json
python scripts/perf_probe.py --volume 256 --viewport 1024 > gpurun_out/probe2/c2.json && cat gpurun_out/probe2/c2.json
