#!/bin/bash
# builds and runs the microbenchmarks of scripts/ubench on the GPU box; outputs under gpurun_out/ubench/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/ubench
for B in ${@:-tcp_coalesce vmem_rate lds_gather valu_rate}; do
  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/$B scripts/ubench/$B.hip > gpurun_out/ubench/$B.build.log 2>&1 || { cat gpurun_out/ubench/$B.build.log; exit 1; }
  timeout -k 10 200 /tmp/$B > gpurun_out/ubench/$B.txt 2>&1 || { echo "$B failed"; cat gpurun_out/ubench/$B.txt; exit 1; }
  cat gpurun_out/ubench/$B.txt
done
