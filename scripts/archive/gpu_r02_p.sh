#!/bin/bash
# round 2, call P: the slab-staged march (vr_hip_set_brick_plane 5): parity first, then per-view timing at the benchmark size
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02p; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "slab_staged or volume_info" > $OUT/tests.log 2>&1; rc=$?; tail -15 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
echo "== slab-staged, trilinear lit"; timeout -k 10 300 python scripts/perf_probe.py --plane 5 || exit 1
echo "== product policy"; timeout -k 10 300 python scripts/perf_probe.py || exit 1
