#!/bin/bash
# round 4, call n: NEAREST column march — the column test under the bounds-checked build, then the product: copies, column, reference-frame parity,
# volume info, random scenes, whole frames at C3 / C4 against the reference's hashes; then NEAREST / TRILINEAR per-view times.  Stops at the first failure.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_n; mkdir -p $O
VR_HIP_LIB=$PWD/build_variants/libvr_hip_bounds.so timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column" > $O/tests_bc.log 2>&1 || { grep -E "bounds check|fault|Abort|assert|Error" $O/tests_bc.log | head; exit 1; }
echo "bounds-checked build: $(tail -1 $O/tests_bc.log)"
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column" > $O/tests0.log 2>&1 || { grep -E "fault|Abort|assert|Error" $O/tests0.log | head; exit 1; }
tail -1 $O/tests0.log
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for samp in nearest trilinear; do
  timeout -k 10 100 python scripts/perf_probe.py --mode nooptims --sampling $samp --reps 6 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
  echo "== $samp"; cut -c100-330 $O/line.json; cat $O/line.json >> $O/probe.log
done
timeout -k 10 100 python scripts/perf_probe.py --mode nooptims --sampling nearest --plane 9 --reps 6 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
echo "== nearest, column windows off"; cut -c100-330 $O/line.json
