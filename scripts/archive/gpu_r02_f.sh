#!/bin/bash
# round 2, call F: run bricks (one 8-byte gather per sample): parity + per-view timing against the quad copies
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02f; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -5 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
echo "== auto (run bricks on oblique views)"; timeout -k 10 300 python scripts/perf_probe.py || exit 1
echo "== run bricks (z) forced"; timeout -k 10 300 python scripts/perf_probe.py --plane 3 || exit 1
echo "== run bricks (y) forced"; timeout -k 10 300 python scripts/perf_probe.py --plane 4 || exit 1
echo "== (x,y) quad copy forced";             timeout -k 10 300 python scripts/perf_probe.py --plane 0 || exit 1
echo "== auto unlit"; timeout -k 10 300 python scripts/perf_probe.py --light 0 || exit 1
echo "== auto default mode"; timeout -k 10 300 python scripts/perf_probe.py --mode default || exit 1
bash scripts/gpu_pmc.sh $OUT/pmc tcc --views 0,1,2,3,4,5,6,7 || exit 1
python scripts/pmc_per_view.py $OUT/pmc 3 | tee $OUT/pmc_per_view.txt
