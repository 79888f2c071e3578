#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03g; mkdir -p $O
for i in 1 2 3; do
python scripts/perf_probe.py --mode default --views 5,3 --each --reps 8 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); print(d["each"])'
done
python scripts/perf_probe.py --mode default --views 5,3 --each --reps 8 --sched 0 2>/dev/null | python -c 'import json,sys; d=json.load(sys.stdin); print("sched0", d["each"])'
