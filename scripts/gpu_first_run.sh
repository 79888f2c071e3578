#!/bin/bash
# smoke + default bench on the GPU box; outputs under gpurun_out/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1 && tail -2 gpurun_out/smoke.log
timeout -k 10 600 python bench.py "$@" > gpurun_out/bench1.json 2> gpurun_out/bench1.err && cat gpurun_out/bench1.json
