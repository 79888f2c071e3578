#!/bin/bash
# round 2, call Y: non-temporal cache policy on the managed gathers (streamed once per workgroup)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
echo "== product"; timeout -k 10 300 python scripts/perf_probe.py || exit 1
echo "== nt gathers"; VR_HIP_LIB=build_variants/libvr_hip_ldnt.so timeout -k 10 300 python scripts/perf_probe.py || exit 1
