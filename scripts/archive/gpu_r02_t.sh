#!/bin/bash
# round 2, call T: 24 instead of 32 waves per CU (16 KiB LDS pad) for the voxel-brick frames of NEAREST as well?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== nearest product"; timeout -k 10 300 python scripts/perf_probe.py --sampling nearest || exit 1
echo "== nearest, voxel bricks padded"; VR_HIP_LIB=build_variants/libvr_hip_padvox.so timeout -k 10 300 python scripts/perf_probe.py --sampling nearest || exit 1
