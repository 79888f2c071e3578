#!/bin/bash
# round 2, call E: -m gpu tier (Q8 mode, stated TRILINEAR tolerances, reference whole-frame hashes) + 8-byte gather microbenchmark
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02e; mkdir -p $OUT gpurun_out/ubench
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.log
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/tg64 scripts/ubench/tcp_gather64.hip
timeout -k 5 200 /tmp/tg64 > gpurun_out/ubench/tcp_gather64.txt 2> gpurun_out/ubench/tcp_gather64.err; echo "ubench rc=$?"
cat gpurun_out/ubench/tcp_gather64.txt; tail -3 gpurun_out/ubench/tcp_gather64.err
