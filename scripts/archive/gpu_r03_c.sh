#!/bin/bash
# round 3, call c: measured-cost tile order — parity, then default / ertonly / full march with the order on and off
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03c; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for sched in 1 0; do for mode in default nooptims; do for s in trilinear nearest; do
  timeout -k 10 200 python scripts/perf_probe.py --mode $mode --sampling $s --sched $sched > $O/probe_${sched}_${mode}_$s.json 2>$O/probe.err || { tail -5 $O/probe.err; exit 1; }
  echo "sched=$sched $(python -c 'import json,sys; d=json.load(open(sys.argv[1])); print(d["mode"], d["sampling"], d["mean_ms"], list(d["kernel_ms_per_view"].values()))' $O/probe_${sched}_${mode}_$s.json)"
done; done; done
