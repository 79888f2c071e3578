#!/bin/bash
# round 2, call AC: 32x32-pixel workgroups (the 1024-thread instantiation with 64-bit z tables) against 32x16 on the same quad copy
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for plane in 0 1 2; do
  echo "== quad plane $plane, 512 threads"; timeout -k 10 300 python scripts/perf_probe.py --plane $plane || exit 1
  echo "== quad plane $plane, 1024 threads"; timeout -k 10 300 python scripts/perf_probe.py --plane $plane --wide 2 || exit 1
done
