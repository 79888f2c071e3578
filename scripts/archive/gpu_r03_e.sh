#!/bin/bash
# round 3, call e: parity tier on the product build, then gpu_r03_d.sh-style A/B. usage: gpu_r03_e.sh <tag> <variant...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
exec bash scripts/gpu_r03_d.sh "$@"
