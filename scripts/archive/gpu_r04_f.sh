#!/bin/bash
# round 4, call f: what bounds the column march — prefetch depth 3 / 5 / 8 / 12, no dense part, no gathers (timing-only builds), views 0, 2, 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_f; mkdir -p $O
for lib in product col_d2 col_d4 col_d5 col_nodense col_noload col_nodense_noload; do
  if [ $lib = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$lib.so; fi
  for light in 0.6 0.0; do
    echo "== $lib light $light" | tee -a $O/probe.log
    timeout -k 10 120 python scripts/perf_probe.py --mode nooptims --views 0,2,3 --light $light --reps 6 2>> $O/probe.err | tee -a $O/probe.log | cut -c100-260
  done
done
