#!/bin/bash
# round 2: bench.py the three ways the driver may start it (direct, through its own launcher, under torch.distributed.run)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_bench; mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -20 $OUT/bench_n1.err; exit 1; }
cat $OUT/bench_n1.json
timeout -k 10 600 python bench.py --force-launcher --no-cpu-baseline --no-extras > $OUT/bench_launcher.json 2> $OUT/bench_launcher.err || { tail -20 $OUT/bench_launcher.err; exit 1; }
cat $OUT/bench_launcher.json
