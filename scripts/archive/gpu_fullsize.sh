#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fullsize
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -s > gpurun_out/fullsize/tests.log 2>&1 || { tail -40 gpurun_out/fullsize/tests.log; exit 1; }
grep -E "config 5|passed|failed" gpurun_out/fullsize/tests.log
