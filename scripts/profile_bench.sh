#!/bin/bash
# rocprofv3 evidence for bench.py on the GPU box: kernel-trace stats + separate PMC passes (FETCH_SIZE, WRITE_SIZE).
# Outputs under gpurun_out/profile_<tag>/ ; copy the summaries you want judged into profiles/.
TAG=${1:-r04}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
ARGS="--steps 16 --warmup 8 --no-cpu-baseline --no-extras $@"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py $ARGS > $OUT/stats_bench.json 2> $OUT/stats.err || { echo stats failed; tail -5 $OUT/stats.err; exit 1; }
cat $OUT/stats_bench.json
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python bench.py $ARGS > $OUT/fetch_bench.json 2> $OUT/fetch.err || { echo fetch failed; tail -5 $OUT/fetch.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python bench.py $ARGS > $OUT/write_bench.json 2> $OUT/write.err || { echo write failed; tail -5 $OUT/write.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/tcc -- python bench.py $ARGS > $OUT/tcc_bench.json 2> $OUT/tcc.err || { echo tcc failed; exit 1; }
python scripts/pmc_summary.py $OUT march_kernel 16:16 | tee $OUT/pmc_summary.txt        # the 16 timed launches (8 set-up frames, one per view, + 8 warm-up frames before them)
python scripts/timed_launches.py $OUT/stats 8 16 8 > $OUT/timed_launches.json; grep -E "timed_mean|timed_max" $OUT/timed_launches.json
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs cat | head -12 | tee $OUT/kernel_stats_head.csv
