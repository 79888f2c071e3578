#!/bin/bash
# wave tile shape (8x8 / 16x4 / 4x16) on the run-brick views: forced lane_map = order + 4 * shape against the automatic choice
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_mapping or trilinear_bit_exact" 2>&1 | tail -1
python scripts/perf_probe.py --views 1,3,4,5,6,7 > $O/auto.json 2>/dev/null; python -c 'import json,sys; d=json.load(open(sys.argv[1])); print("auto", d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/auto.json
for lm in 0 4 8 1 5 9; do
python scripts/perf_probe.py --views 1,3,4,5,6,7 --tile-map $lm,0,0 > $O/lm$lm.json 2>/dev/null || exit 1
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print("lane_map", sys.argv[2], d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/lm$lm.json $lm
done
