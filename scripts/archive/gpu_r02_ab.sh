#!/bin/bash
# round 2, call AB: what the second gather of a quad-brick sample costs the aligned views (timing only: wrong images)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== product, quad copies per view (plane -1), views 0 2"; timeout -k 10 300 python scripts/perf_probe.py --views 0,2 || exit 1
echo "== one gather per sample"; VR_HIP_LIB=build_variants/libvr_hip_oneload.so timeout -k 10 300 python scripts/perf_probe.py --views 0,2 || exit 1
