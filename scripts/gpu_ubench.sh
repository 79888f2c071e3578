#!/bin/bash
# builds and runs the microbenchmarks of scripts/ubench on the GPU box; outputs under gpurun_out/ubench/
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ubench
for B in ${@:-tcp_coalesce vmem_rate lds_gather valu_rate}; do
  hipcc --offload-arch=gfx950 -O3 -o /tmp/$B scripts/ubench/$B.hip 2> /dev/null
  timeout -k 10 120 /tmp/$B > gpurun_out/ubench/$B.txt
  cat gpurun_out/ubench/$B.txt
done
