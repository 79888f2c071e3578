#!/bin/bash
# round 2, call H: free-running march (padded tables, finished lanes stop), scaled-domain NEAREST: parity + timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02h; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -8 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
for seed in 11 12 13; do VR_TEST_SEED=$seed timeout -k 10 600 python -m pytest tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -1; done
echo "== trilinear lit"; timeout -k 10 300 python scripts/perf_probe.py || exit 1
echo "== nearest lit"; timeout -k 10 300 python scripts/perf_probe.py --sampling nearest || exit 1
echo "== trilinear default"; timeout -k 10 300 python scripts/perf_probe.py --mode default || exit 1
echo "== nearest default"; timeout -k 10 300 python scripts/perf_probe.py --mode default --sampling nearest || exit 1
