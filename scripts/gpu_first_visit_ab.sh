#!/bin/bash
# A/B of the predicted launch order for first frames (VR_TILE_ESTIMATE=0 / 1): the placement tests, bench.py's first_visit legs, and the
# durations of the two kernels that run in front of such a frame (rocprofv3 kernel trace of a default-mode bench run).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fv; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q -k "tile_scheduling or orders_and_recordings or random" > $O/tests.log 2>&1 || { tail -25 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for e in 0 1; do
  VR_TILE_ESTIMATE=$e timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_$e.json 2> $O/bench_$e.err || { tail -5 $O/bench_$e.err; exit 1; }
  python - $O/bench_$e.json $e <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
fv = d["extras"]["first_visit"]
for k in ("default_trilinear", "default_nearest"):
    print("estimate", sys.argv[2], k, {x: fv[k][x] for x in ("protocol_first_ms", "protocol_first_ms_max", "protocol_steady_ms", "protocol_first_over_steady", "moving_ms", "moving_steady_ms")}, fv[k]["protocol_first_per_view_ms"])
PY
done
rm -rf $O/stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --mode default --steps 16 --warmup 8 --no-cpu-baseline --no-extras > $O/bench_default.json 2> $O/stats.err || { tail -5 $O/stats.err; exit 1; }
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); grep -i "tile_estimate\|tile_order" $f | cut -c1-200 | tee $O/pre_kernels.txt
