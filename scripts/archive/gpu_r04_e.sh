#!/bin/bash
# round 4, call e: column march — copies byte for byte, parity (dedicated test, layouts, random scenes), then per-view timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_e; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_copies.py tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q -k "copies or column or layouts_agree or volume_info or random or nearest_bit_exact" 2>&1 | tee $O/tests.log | tail -25
grep -q "failed\|error" $O/tests.log && exit 1
for plane in -1 9; do
  echo "== full march, plane $plane" | tee -a $O/probe.log
  timeout -k 10 200 python scripts/perf_probe.py --mode nooptims --plane $plane --reps 6 2>> $O/probe.err | tee -a $O/probe.log | cut -c1-600
done
