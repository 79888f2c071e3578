#!/bin/bash
# A/B of compile-time kernel variants on the GPU box: rebuilds the library per variant (EXTRA=...) and runs the probe.
# usage: gpu_variants.sh "<probe args>" "<EXTRA flags 1>" "<EXTRA flags 2>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/variants
PROBE="$1"; shift
for V in "$@"; do
  rm -f volume-rendering_amd/libvr_hip.so
  make -C volume-rendering_amd/csrc EXTRA="$V" > gpurun_out/variants/build.log 2>&1 || { tail -5 gpurun_out/variants/build.log; exit 1; }
  echo "== EXTRA='$V'"
  python scripts/perf_probe.py $PROBE || exit 1
done
