#!/bin/bash
# measurement: chunk plane (x,z) and (y,z) brick orders under each lane order (potential of per-view copies)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in "-DVR_BRICK_BITS=0,2,5,4,7,8,1,3,6" "-DVR_BRICK_BITS=4,7,8,0,2,5,1,3,6"; do
  rm -f volume-rendering_amd/libvr_hip.so
  make -C volume-rendering_amd/csrc EXTRA="$V" > /dev/null 2>&1 || exit 1
  for TM in "" "0,0,0" "1,0,0" "2,0,0" "2,1,1" "2,1,0" "2,0,1"; do
    echo "== EXTRA='$V' tile-map '$TM'"
    python scripts/perf_probe.py --light 0 --reps 2 ${TM:+--tile-map $TM} || exit 1
  done
done
