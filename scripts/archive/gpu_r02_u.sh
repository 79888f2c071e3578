#!/bin/bash
# round 2, call U: managed loads for TRILINEAR with 2-byte voxels: parity tier, then timing at 1024^3 u16
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02u; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -5 $OUT/tests.log; [ $rc -eq 0 ] || exit $rc
echo "== 1024^3 u16 @ 2048^2 trilinear"; timeout -k 10 300 python scripts/perf_probe.py --bpv 2 --reps 3 || exit 1
echo "== 1024^3 u16 @ 2048^2 trilinear, default mode"; timeout -k 10 300 python scripts/perf_probe.py --bpv 2 --reps 3 --mode default || exit 1
echo "== 512^3 u16 @ 1024^2 trilinear (32-bit offsets)"; timeout -k 10 300 python scripts/perf_probe.py --bpv 2 --reps 3 --volume 512 --viewport 1024 || exit 1
