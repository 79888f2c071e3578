#!/bin/bash
# round 2, call G: copy selection policy v2 (run bricks for oblique, unalignable orthogonal and x/y perspective views)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02g; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -5 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
echo "== auto lit"; timeout -k 10 300 python scripts/perf_probe.py || exit 1
echo "== auto q8"; timeout -k 10 300 python scripts/perf_probe.py --sampling q8 || exit 1
echo "== auto 512 @ 1080 lit"; timeout -k 10 300 python scripts/perf_probe.py --volume 512 --viewport 1080 || exit 1
echo "== auto 256 @ 1024 lit"; timeout -k 10 300 python scripts/perf_probe.py --volume 256 --viewport 1024 || exit 1
