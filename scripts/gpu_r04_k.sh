#!/bin/bash
# round 4, call k: the evidence run — whole GPU tier, smoke, bench.py (full line), rocprofv3 passes of bench.py, per-view counters of the
# full march (1-byte voxels) and of the 2-byte workload (1024^3 u16 @ 2048^2).  Stops at the first step that fails.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_k; mkdir -p $O
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_k/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'per view', d['roofline'].get('per_view_kernel_ms'))
PY
bash scripts/profile_bench.sh r04 > $O/profile.log 2>&1 || { tail -20 $O/profile.log; exit 1; }
grep -E "timed_mean|FETCH_SIZE|WRITE_SIZE" $O/profile.log
bash scripts/gpu_pmc.sh gpurun_out/r04_pmc_u8 sq1,sq2,tcc,fetch,tcp1 --mode nooptims --views 0,1,2,3,4,5,6,7 > $O/pmc_u8.log 2>&1 || { tail -10 $O/pmc_u8.log; exit 1; }
python scripts/pmc_per_view.py gpurun_out/r04_pmc_u8 6 march_kernel 2 > $O/per_view_pmc_u8.txt; cat $O/per_view_pmc_u8.txt
python scripts/valu_busy.py $O/per_view_pmc_u8.txt > $O/valu_busy_u8.txt; cat $O/valu_busy_u8.txt
timeout -k 10 150 python scripts/perf_probe.py --mode nooptims --bpv 2 --reps 4 > $O/u16_ms.json 2>> $O/probe.err || { tail -5 $O/probe.err; exit 1; }
cut -c1-400 $O/u16_ms.json
bash scripts/gpu_pmc.sh gpurun_out/r04_pmc_u16 sq2,tcc,fetch,tcp1 --mode nooptims --bpv 2 --views 0,1,2,3,4,5,6,7 > $O/pmc_u16.log 2>&1 || { tail -10 $O/pmc_u16.log; exit 1; }
python scripts/pmc_per_view.py gpurun_out/r04_pmc_u16 6 march_kernel 2 > $O/per_view_pmc_u16.txt; cat $O/per_view_pmc_u16.txt
