#!/bin/bash
# round 3, call d: A/B of library variants (build_variants/libvr_hip_<name>.so) — per-view times. usage: gpu_r03_d.sh <tag> <variant...>
# env MODES / SAMPLINGS / PROBE_ARGS select what is probed
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for v in product "$@"; do
  if [ $v = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$v.so; fi
  for mode in ${MODES:-nooptims default}; do for s in ${SAMPLINGS:-trilinear}; do
    timeout -k 10 200 python scripts/perf_probe.py --mode $mode --sampling $s $PROBE_ARGS > $O/probe_${v}_${mode}_$s.json 2>$O/probe.err || { tail -5 $O/probe.err; exit 1; }
    echo "$v $(python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(d["mode"], d["sampling"], d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/probe_${v}_${mode}_$s.json)"
  done; done
done
