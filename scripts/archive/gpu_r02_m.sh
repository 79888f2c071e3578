#!/bin/bash
# round 2, call M: voxel bricks for NEAREST: parity + timing against the (x,y) quad copy
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02m; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -5 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
for seed in 31 32; do VR_TEST_SEED=$seed timeout -k 10 600 python -m pytest tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -1; done
echo "== nearest, voxel bricks"; timeout -k 10 300 python scripts/perf_probe.py --sampling nearest || exit 1
echo "== nearest, quad copy forced"; timeout -k 10 300 python scripts/perf_probe.py --sampling nearest --plane 0 || exit 1
echo "== nearest default mode, voxel bricks"; timeout -k 10 300 python scripts/perf_probe.py --sampling nearest --mode default || exit 1
echo "== trilinear"; timeout -k 10 300 python scripts/perf_probe.py || exit 1
