#!/bin/bash
# tests + per-view timings: automatic tile mapping against forced ones
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tilemap
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/tilemap/tests.log 2>&1 || { tail -30 gpurun_out/tilemap/tests.log; exit 1; }
tail -2 gpurun_out/tilemap/tests.log
echo "== unlit auto"; python scripts/perf_probe.py --light 0
echo "== lit auto"; python scripts/perf_probe.py
echo "== lit rows"; python scripts/perf_probe.py --tile-map 0,0,0
echo "== c2 auto"; python scripts/perf_probe.py --volume 256 --viewport 1024
echo "== c2 rows"; python scripts/perf_probe.py --volume 256 --viewport 1024 --tile-map 0,0,0
echo "== 512@1080 auto"; python scripts/perf_probe.py --volume 512 --viewport 1080
echo "== 512@1080 rows"; python scripts/perf_probe.py --volume 512 --viewport 1080 --tile-map 0,0,0
