#!/bin/bash
# round 2, call R: where configuration 5 (2048^3 u16 @ 4096^2) and the u16 path stand
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== 1024^3 u16 @ 2048^2 trilinear"; timeout -k 10 300 python scripts/perf_probe.py --bpv 2 --reps 3 || exit 1
echo "== 1024^3 u16 @ 2048^2 nearest"; timeout -k 10 300 python scripts/perf_probe.py --bpv 2 --reps 3 --sampling nearest || exit 1
echo "== 2048^3 u16 @ 4096^2 trilinear"; timeout -k 10 600 python scripts/perf_probe.py --volume 2048 --viewport 4096 --bpv 2 --reps 2 || exit 1
echo "== 2048^3 u16 @ 4096^2 nearest"; timeout -k 10 600 python scripts/perf_probe.py --volume 2048 --viewport 4096 --bpv 2 --reps 2 --sampling nearest || exit 1
