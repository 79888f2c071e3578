#!/bin/bash
# round 3, call k: the pixel-block march — parity tier, then per-view full-march times with it on (product) and off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for s in trilinear q8; do
timeout -k 10 200 python scripts/perf_probe.py --mode nooptims --sampling $s > $O/probe_$s.json 2>$O/probe.err || { tail -5 $O/probe.err; exit 1; }
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(d["sampling"], d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/probe_$s.json
done
