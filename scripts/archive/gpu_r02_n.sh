#!/bin/bash
# round 2, call N: does the 16 KiB LDS pad of run-brick launches cost the short-ray modes (default = ESL + ERT, ERT only)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "" build_variants/libvr_hip_nopad.so; do
  for mode in default ertonly; do
    echo "== lib=${lib:-product} mode=$mode"; VR_HIP_LIB=$lib timeout -k 10 300 python scripts/perf_probe.py --mode $mode || exit 1
  done
done
