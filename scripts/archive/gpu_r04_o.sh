#!/bin/bash
# round 4, call o: early termination without leaping through the column kernels (VR_COL_ERT=1 / 0): parity subset, then per-view times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_o; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for ert in 1 0; do for samp in trilinear nearest; do
  VR_COL_ERT=$ert timeout -k 10 100 python scripts/perf_probe.py --mode ertonly --sampling $samp --reps 6 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
  echo "== VR_COL_ERT=$ert $samp"; cut -c100-330 $O/line.json
done; done
