#!/bin/bash
# round 2, call O: upper bound of an LDS-staged march: the run gather replaced by a 64-bit LDS read (no staging, zero data)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== product, run bricks (z) forced"; timeout -k 10 300 python scripts/perf_probe.py --plane 3 || exit 1
echo "== LDS gather, run bricks (z) forced"; VR_HIP_LIB=build_variants/libvr_hip_ldsg.so timeout -k 10 300 python scripts/perf_probe.py --plane 3 || exit 1
echo "== LDS gather, run bricks (z) forced, unlit"; VR_HIP_LIB=build_variants/libvr_hip_ldsg.so timeout -k 10 300 python scripts/perf_probe.py --plane 3 --light 0 || exit 1
