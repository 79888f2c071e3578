#!/bin/bash
# round 2, call Z: rocprofv3 kernel stats of bench.py in NEAREST sampling (the reference-pinned mode) and in the default mode
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile_r02_modes; mkdir -p $OUT
for cfg in "nearest nooptims" "trilinear default" "nearest default"; do
  set -- $cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$1_$2 -- python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-extras --sampling $1 --mode $2 > $OUT/$1_$2.json 2> $OUT/$1_$2.err || { tail -5 $OUT/$1_$2.err; exit 1; }
  cat $OUT/$1_$2.json | cut -c1-200
  find $OUT/$1_$2 -name "*kernel_stats.csv" | head -1 | xargs grep raymarch | tee $OUT/$1_$2_raymarch.csv
done
