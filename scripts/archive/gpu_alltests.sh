#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/alltests
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -s > gpurun_out/alltests/tests.log 2>&1 || { tail -40 gpurun_out/alltests/tests.log; exit 1; }
grep -E "config 5|passed|failed" gpurun_out/alltests/tests.log
