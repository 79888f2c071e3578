#!/bin/bash
# round 3, call b: A/B of kernel variants built in the container (build_variants/libvr_hip_<name>.so): parity tier on the product
# build, then per-view times of the full march and the default mode for every variant.  usage: gpu_r03_b.sh <tag> <variant names...>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for v in product "$@"; do
  if [ $v = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$v.so; fi
  for mode in ${MODES:-nooptims default}; do for s in ${SAMPLINGS:-trilinear}; do
    timeout -k 10 200 python scripts/perf_probe.py --mode $mode --sampling $s > $O/probe_${v}_${mode}_$s.json 2>$O/probe_${v}_${mode}_$s.err || { tail -5 $O/probe_${v}_${mode}_$s.err; exit 1; }
    echo "$v $(cat $O/probe_${v}_${mode}_$s.json | python -c 'import json,sys; d=json.load(sys.stdin); print(d["mode"], d["sampling"], d["mean_ms"], list(d["kernel_ms_per_view"].values()))')"
  done; done
done
