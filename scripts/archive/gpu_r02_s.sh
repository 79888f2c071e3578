#!/bin/bash
# round 2, call S: lane order (which pixels of a 4x4 block share a lane quad) on the run-brick / voxel-brick views, forced against automatic
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for samp in trilinear nearest; do
  echo "== $samp automatic"; timeout -k 10 300 python scripts/perf_probe.py --sampling $samp --views 1,3,4,5,6,7 || exit 1
  for lm in 0 1 2; do echo "== $samp lane_map $lm"; timeout -k 10 300 python scripts/perf_probe.py --sampling $samp --views 1,3,4,5,6,7 --tile-map $lm,0,0 || exit 1; done
done
