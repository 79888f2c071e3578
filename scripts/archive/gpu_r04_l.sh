#!/bin/bash
# round 4, call l: empty-space leaping as a pass of its own — whole GPU tier, then the default mode per view (TRILINEAR / NEAREST, tile order on / off)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_l; mkdir -p $O
timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for samp in trilinear nearest; do for sched in 1 0; do
  echo "== default $samp sched $sched" | tee -a $O/probe.log
  timeout -k 10 100 python scripts/perf_probe.py --mode default --sampling $samp --sched $sched --reps 8 --each 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
  cat $O/line.json >> $O/probe.log
  python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_l/line.json'))
print('mean', d['mean_ms'], 'per view', [d['kernel_ms_per_view'][k] for k in sorted(d['kernel_ms_per_view'])], 'view5 min/max', min(d['each']['5']), max(d['each']['5']), 'view3', min(d['each']['3']), max(d['each']['3']))
PY
done; done
timeout -k 10 100 python scripts/perf_probe.py --mode nooptims --reps 6 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
cut -c100-330 $O/line.json
