#!/bin/bash
# A/B of the predicted launch order for first frames (VR_TILE_ESTIMATE=0 / 1): the placement test, then bench.py's first_visit legs.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/fv; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile_scheduling or orders_and_recordings" > $O/tests.log 2>&1 || { tail -25 $O/tests.log; exit 1; }
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
