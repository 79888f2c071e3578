#!/bin/bash
# round 3: (a) how many frames in flight a rank of an N-rank run wants (1..4 streams, band sets of N = 2, 4, 8);
# (b) host-side cost of a bench.py step through the rank path (launcher, RCCL gather to self) on the small configuration
set -e
mkdir -p gpurun_out/r03v
python scripts/overlap_probe.py --ranks 2,4,8 --streams 1,2,3,4 > gpurun_out/r03v/overlap.json
python scripts/overlap_probe.py --ranks 4,8 --streams 1,2,3,4 --mode default > gpurun_out/r03v/overlap_default.json
python bench.py --force-launcher --config c2 --steps 400 --warmup 40 --no-extras --no-cpu-baseline > gpurun_out/r03v/host_c2_one.json
VR_BENCH_TWO_STREAMS=1 python bench.py --force-launcher --config c2 --steps 400 --warmup 40 --no-extras --no-cpu-baseline > gpurun_out/r03v/host_c2_two.json
python bench.py --config c2 --steps 400 --warmup 40 --no-extras --no-cpu-baseline > gpurun_out/r03v/host_c2_direct.json
