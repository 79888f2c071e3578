#!/bin/bash
# oct bricks for 2-byte voxels: the whole -m gpu tier (config 5 included), then 1024^3 u16 @ 2048^2 and config 5 per view
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03r; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
grep -E "config 5|passed|failed" $O/tests.log
for plane in -1 0; do
timeout -k 10 300 python scripts/perf_probe.py --bpv 2 --plane $plane > $O/u16_$plane.json 2>$O/probe.err || { tail -5 $O/probe.err; exit 1; }
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print("1024^3 u16 plane", sys.argv[2], d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/u16_$plane.json $plane
done
timeout -k 10 600 python scripts/perf_probe.py --bpv 2 --volume 2048 --viewport 4096 --reps 2 > $O/c5.json 2>$O/probe.err || { tail -5 $O/probe.err; exit 1; }
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print("config 5", d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/c5.json
timeout -k 10 600 python scripts/perf_probe.py --bpv 2 --volume 2048 --viewport 4096 --reps 2 --plane 0 > $O/c5q.json 2>$O/probe.err || { tail -5 $O/probe.err; exit 1; }
python -c 'import json,sys; d=json.load(open(sys.argv[1])); print("config 5 quad", d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/c5q.json
