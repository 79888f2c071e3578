#!/bin/bash
# round 3 evidence at HEAD: the -m gpu tier + smoke, the bench.py line, rocprofv3 kernel stats + PMC traffic of the same command, per-view
# kernel times (full march lit / unlit / NEAREST, default mode), per-view counters of all 8 views for the full march and the default mode
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03_final; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
timeout -k 10 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
tail -1 $OUT/smoke.log
timeout -k 10 600 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err || { tail -20 $OUT/bench_n1.err; exit 1; }
python -c "import json; d=json.load(open('$OUT/bench_n1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['per_view_kernel_ms'])"
bash scripts/profile_bench.sh r03 || exit 1
echo "== per-view timing"
python scripts/perf_probe.py > $OUT/per_view_trilinear.json && cat $OUT/per_view_trilinear.json
python scripts/perf_probe.py --sampling nearest > $OUT/per_view_nearest.json && cat $OUT/per_view_nearest.json
python scripts/perf_probe.py --light 0 > $OUT/per_view_trilinear_unlit.json && cat $OUT/per_view_trilinear_unlit.json
python scripts/perf_probe.py --mode default > $OUT/per_view_default_trilinear.json && cat $OUT/per_view_default_trilinear.json
python scripts/perf_probe.py --mode default --sampling nearest > $OUT/per_view_default_nearest.json && cat $OUT/per_view_default_nearest.json
python scripts/perf_probe.py --mode default --sched 0 > $OUT/per_view_default_trilinear_sched0.json && cat $OUT/per_view_default_trilinear_sched0.json
bash scripts/gpu_pmc.sh $OUT/pmc sq1,sq2,tcp1,tcc,fetch --views 0,1,2,3,4,5,6,7 || exit 1
python scripts/pmc_per_view.py $OUT/pmc 6 raymarch 2 > $OUT/pmc_per_view.txt; cat $OUT/pmc_per_view.txt
bash scripts/gpu_pmc.sh $OUT/pmc_default sq1,sq2,tcc,fetch --mode default --views 0,1,2,3,4,5,6,7 || exit 1
python scripts/pmc_per_view.py $OUT/pmc_default 6 raymarch 2 > $OUT/pmc_default_per_view.txt; cat $OUT/pmc_default_per_view.txt
find gpurun_out/profile_r03/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_full.csv
