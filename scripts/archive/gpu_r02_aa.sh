#!/bin/bash
# round 2, call AA: table lookups of axes the rays do not move along hoisted out of the sample chain (exactly axis-aligned orthogonal views)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02aa; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -8 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
echo "== trilinear"; timeout -k 10 300 python scripts/perf_probe.py || exit 1
echo "== trilinear default mode"; timeout -k 10 300 python scripts/perf_probe.py --mode default || exit 1
