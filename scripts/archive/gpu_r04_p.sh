#!/bin/bash
# round 4, call p: cooperative look-ahead inside the march kernel — parity of everything that leaps, then the default mode per view for build variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_p; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for lib in product "$@"; do
  if [ $lib = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$lib.so; fi
  for samp in trilinear nearest; do for sched in 1 0; do
    timeout -k 10 100 python scripts/perf_probe.py --mode default --sampling $samp --sched $sched --reps 8 --each 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
    python - "$lib" "$samp" "$sched" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04_p/line.json'))
print(sys.argv[1], sys.argv[2], 'sched', sys.argv[3], 'mean', d['mean_ms'], 'per view', [d['kernel_ms_per_view'][k] for k in sorted(d['kernel_ms_per_view'])], 'v5', min(d['each']['5']), max(d['each']['5']))
PY
  done; done
done
