#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03u; mkdir -p $O
python scripts/perf_probe.py --sampling nearest > $O/auto.json 2>/dev/null; python -c 'import json,sys; d=json.load(open(sys.argv[1])); print("nearest auto", d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/auto.json
for lm in 0 4 8 1 5 9 2 6 10; do
python scripts/perf_probe.py --sampling nearest --tile-map $lm,0,0 > $O/lm$lm.json 2>/dev/null || exit 1
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print("nearest lane_map", sys.argv[2], d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/lm$lm.json $lm
done
for lm in 2 6 10; do
python scripts/perf_probe.py --views 0,2 --tile-map $lm,0,0 > $O/t$lm.json 2>/dev/null || exit 1
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print("trilinear views 0,2 lane_map", sys.argv[2], d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/t$lm.json $lm
done
