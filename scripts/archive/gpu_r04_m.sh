#!/bin/bash
# round 4, call m: look-ahead threshold of the leap pass — default mode per view for build variants
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_m; mkdir -p $O
for lib in product "$@"; do
  if [ $lib = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$lib.so; fi
  for samp in trilinear nearest; do
    timeout -k 10 100 python scripts/perf_probe.py --mode default --sampling $samp --sched 1 --reps 8 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
    python - "$lib" "$samp" <<'PY'
import json,sys
d=json.load(open('gpurun_out/r04_m/line.json'))
print(sys.argv[1], sys.argv[2], 'mean', d['mean_ms'], 'per view', [d['kernel_ms_per_view'][k] for k in sorted(d['kernel_ms_per_view'])])
PY
  done
done
