#!/bin/bash
# round 2 evidence at HEAD: bench.py line, rocprofv3 kernel stats + PMC traffic of the same command, per-view counters of all 8 views
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02_final; mkdir -p $OUT
timeout -k 10 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -20 $OUT/bench_n1.err; exit 1; }
cat $OUT/bench_n1.json
bash scripts/profile_bench.sh r02 || exit 1
echo "== per-view timing"; python scripts/perf_probe.py | tee $OUT/per_view_trilinear.json
python scripts/perf_probe.py --sampling nearest | tee $OUT/per_view_nearest.json
python scripts/perf_probe.py --light 0 | tee $OUT/per_view_trilinear_unlit.json
bash scripts/gpu_pmc.sh $OUT/pmc sq1,sq2,tcp1,tcc,fetch --views 0,1,2,3,4,5,6,7 || exit 1
python scripts/pmc_per_view.py $OUT/pmc 3 | tee $OUT/pmc_per_view.txt
find gpurun_out/profile_r02/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_full.csv
