#!/bin/bash
# tests + per-view timings: brick copy per view against each forced chunk plane
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/planes
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/planes/tests.log 2>&1 || { tail -30 gpurun_out/planes/tests.log; exit 1; }
tail -2 gpurun_out/planes/tests.log
for PL in -1 0 1 2; do
  echo "== unlit plane $PL"; python scripts/perf_probe.py --plane $PL --reps 3 --light 0
done
echo "== lit auto"; python scripts/perf_probe.py
echo "== default auto"; python scripts/perf_probe.py --mode default
echo "== c2 auto"; python scripts/perf_probe.py --volume 256 --viewport 1024
echo "== nearest"; python scripts/perf_probe.py --sampling nearest
