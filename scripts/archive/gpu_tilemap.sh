#!/bin/bash
# tests + per-view timings: automatic tile mapping against forced ones
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tilemap
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/tilemap/tests.log 2>&1 || { tail -30 gpurun_out/tilemap/tests.log; exit 1; }
tail -2 gpurun_out/tilemap/tests.log
for TM in "" "0,0,0" "2,0,0" "2,1,1" "1,0,0" "0,1,0"; do
  echo "== nearest tile-map '$TM'"
  python scripts/perf_probe.py --sampling nearest --reps 3 ${TM:+--tile-map $TM}
done
echo "== nearest default auto"; python scripts/perf_probe.py --mode default --sampling nearest
