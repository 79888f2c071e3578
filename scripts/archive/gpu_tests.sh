#!/bin/bash
# the -m gpu tier + smoke; log under gpurun_out/tests/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tests
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/tests/tests.log 2>&1; rc=$?
tail -25 gpurun_out/tests/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
