#!/bin/bash
# round 4, call b: cooperative ESL look-ahead + analytic run-copy choice. parity on the product (VR_COOP_LANES=8), then default-mode per-view
# times with the tile order on (sched 1, steady state) and off (sched 0 = what a first frame gets) for coop 0 / 8 / 64; view 1 analytic vs measured
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_b; mkdir -p $O
timeout -k 10 240 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_copies.py -m gpu -x -v 2>&1 | tee $O/tests.log | grep -E "PASSED|FAILED|ERROR|passed|failed" || { tail -40 $O/tests.log; exit 1; }
grep -q " passed" $O/tests.log || exit 1
grep -q "failed" $O/tests.log && exit 1
for lib in product coop0 coop64; do
  if [ $lib = product ]; then unset VR_HIP_LIB; else export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$lib.so; fi
  for samp in trilinear nearest; do
    for sched in 1 0; do
      echo "== $lib $samp sched $sched" | tee -a $O/probe.log
      timeout -k 10 120 python scripts/perf_probe.py --mode default --sampling $samp --sched $sched --reps 8 --each >> $O/probe.log 2>> $O/probe.err || { tail -5 $O/probe.err; exit 1; }
    done
  done
done
unset VR_HIP_LIB
python - <<'PY'
import json
cur=None
for line in open('gpurun_out/r04_b/probe.log'):
    if line.startswith('=='): cur=line.strip(); continue
    d=json.loads(line)
    each=d['each']
    print(cur, 'mean', d['mean_ms'], 'per view', [d['kernel_ms_per_view'][k] for k in sorted(d['kernel_ms_per_view'])], 'view5 min/max', min(each['5']), max(each['5']))
PY
for plane in -1 6 3 4; do
  echo "== full march views 1,5 plane $plane" | tee -a $O/dual.log
  timeout -k 10 120 python scripts/perf_probe.py --mode nooptims --views 1,5 --plane $plane --reps 8 2>> $O/probe.err | tee -a $O/dual.log | cut -c1-400
done
timeout -k 10 120 python scripts/copy_build_probe.py 1024 1 3 > $O/probe_1024_u8.json 2>> $O/probe.err && python -c "
import json; d=json.load(open('$O/probe_1024_u8.json')); print('generate', d['generate'], d['generate_roofline'])"
timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -v -k "c4 or config3 or partition" 2>&1 | tee $O/c4.log | grep -E "PASSED|FAILED|ERROR|passed|failed"
