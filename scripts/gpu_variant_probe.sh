#!/bin/bash
# A/B of build variants: the column-march parity test with each variant (VR_HIP_LIB), then its kernel times on the three axis-aligned poses.  usage: gpu_variant_probe.sh <variant>...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_i; mkdir -p $O
for lib in "$@"; do
  export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$lib.so
  echo "== $lib"
  timeout -k 10 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column" > $O/tests_$lib.log 2>&1 || { grep -E "fault|Abort|assert|Error" $O/tests_$lib.log | head -5; exit 1; }
  tail -1 $O/tests_$lib.log
  timeout -k 10 60 python scripts/perf_probe.py --mode nooptims --views 0,2,3 --light 0.6 --reps 6 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
  cut -c100-230 $O/line.json
done
