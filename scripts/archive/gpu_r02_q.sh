#!/bin/bash
# round 2, call Q: v_add3_u32 in the quad / voxel address chains: parity tier + per-view timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02q; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -5 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
echo "== trilinear"; timeout -k 10 300 python scripts/perf_probe.py || exit 1
echo "== nearest"; timeout -k 10 300 python scripts/perf_probe.py --sampling nearest || exit 1
