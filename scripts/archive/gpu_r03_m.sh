#!/bin/bash
# pixel-block march: parity tier, then view 0 lit / unlit with it on and off, and the 8-view line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for v in product "$@"; do
if [ $v = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$v.so; fi
for block in 1 0; do for light in 0.6 0; do
timeout -k 10 200 python scripts/perf_probe.py --mode nooptims --views 0 --block $block --light $light > $O/p.json 2>$O/probe.err || { tail -5 $O/probe.err; exit 1; }
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], sys.argv[3], sys.argv[4], d["mean_ms"])' $O/p.json $v block=$block light=$light
done; done; done
unset VR_HIP_LIB
timeout -k 10 200 python scripts/perf_probe.py --mode nooptims > $O/all.json 2>$O/probe.err || { tail -5 $O/probe.err; exit 1; }
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/all.json
