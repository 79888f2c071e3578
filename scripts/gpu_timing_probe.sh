#!/bin/bash
# Kernel times of timing-only build variants (no parity test: such builds render wrong images on purpose).  usage: gpu_timing_probe.sh "<perf_probe args>" <variant>...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_t; mkdir -p $O
ARGS="$1"; shift
for lib in "$@"; do
  export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$lib.so
  echo "== $lib"
  timeout -k 10 90 python scripts/perf_probe.py $ARGS 2>> $O/probe.err > $O/line_$lib.json || { tail -5 $O/probe.err; exit 1; }
  python - $O/line_$lib.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print({k: round(v, 3) for k, v in d["kernel_ms_per_view"].items()}, "mean", round(d["mean_ms"], 3))
PY
done
