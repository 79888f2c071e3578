#!/bin/bash
# round 2, call V: 16 waves per CU (40 KB pad) against 24 (16 KB, product) for the run-brick frames, per view
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== product (24 waves on run-brick frames)"; timeout -k 10 300 python scripts/perf_probe.py || exit 1
echo "== 16 waves on run-brick frames"; VR_HIP_LIB=build_variants/libvr_hip_pad2wg.so timeout -k 10 300 python scripts/perf_probe.py || exit 1
