#!/bin/bash
# bench.py under torch.distributed.run with ONE rank: exercises the RCCL init / gather / barrier path on the 1-GPU box
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/dist1
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 16 --warmup 4 --no-cpu-baseline --no-extras > gpurun_out/dist1/bench.json 2> gpurun_out/dist1/bench.err || { tail -30 gpurun_out/dist1/bench.err; exit 1; }
cat gpurun_out/dist1/bench.json
