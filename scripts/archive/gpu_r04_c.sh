#!/bin/bash
# round 4, call c: which run copy per entry face — the 8 rules (VR_DUAL_RULE bit f: entry through a face of axis f reads the copy along y)
# against the measured per-block choice (plane 6) and both single copies, on three oblique orthogonal poses; generator timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_c; mkdir -p $O
POSES="-45,-45,0;30,20,0;-60,35,10;20,-70,0"
for plane in 6 3 4; do
  echo "== plane $plane" | tee -a $O/dual.log
  timeout -k 10 120 python scripts/perf_probe.py --mode nooptims --pose="$POSES" --plane $plane --reps 6 2>> $O/err.log | tee -a $O/dual.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_view'])"
done
for rule in 0 1 2 3 4 5 6 7; do
  echo "== rule $rule" | tee -a $O/dual.log
  VR_DUAL_RULE=$rule timeout -k 10 120 python scripts/perf_probe.py --mode nooptims --pose="$POSES" --plane -1 --reps 6 2>> $O/err.log | tee -a $O/dual.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms_per_view'])"
done
timeout -k 10 900 python -m pytest tests/test_gpu_bounds.py -m gpu -x -v 2>&1 | tee $O/bounds.log | grep -E "PASSED|FAILED|ERROR|passed|failed|rror"
