#!/bin/bash
# full GPU tier: every gpu-marked test, smoke, default bench
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/full
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/full/tests.log 2>&1 || { tail -40 gpurun_out/full/tests.log; exit 1; }
tail -3 gpurun_out/full/tests.log
python __graft_entry__.py smoke > gpurun_out/full/smoke.log 2>&1 && tail -1 gpurun_out/full/smoke.log
timeout -k 10 600 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err && cat gpurun_out/full/bench.json
