#!/bin/bash
# round 4, call h: timing of build variants on views 0,2,3 (full march). usage: gpu_r04_h.sh <light> <variant>...   Stops at the first failure.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_h; mkdir -p $O
L=$1; shift
for lib in product "$@"; do
  if [ $lib = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$lib.so; fi
  for light in $L; do
    echo "== $lib light $light" | tee -a $O/probe.log
    timeout -k 10 60 python scripts/perf_probe.py --mode nooptims --views 0,2,3 --light $light --reps 6 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
    cat $O/line.json >> $O/probe.log; cut -c100-230 $O/line.json
  done
done
