#!/bin/bash
# round 3, call a: the -m gpu tier after the lazy-copy refactor (incl. the 64 new whole-frame TRILINEAR hashes), per-view times of
# the full march and of the default mode (ESL + ERT), and per-view counters of the default mode
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
grep -E "config 5|passed|failed" $O/tests.log
for mode in nooptims default; do for s in trilinear nearest; do
  timeout -k 10 200 python scripts/perf_probe.py --mode $mode --sampling $s > $O/probe_${mode}_$s.json 2>$O/probe_${mode}_$s.err || { tail -5 $O/probe_${mode}_$s.err; exit 1; }
  cat $O/probe_${mode}_$s.json
done; done
bash scripts/gpu_pmc.sh $O/pmc_default sq1,sq2,tcc,fetch --mode default --sampling trilinear --views 0,1,2,3,4,5,6,7 || exit 1
python scripts/pmc_per_view.py $O/pmc_default 3 > $O/pmc_default_per_view.txt; cat $O/pmc_default_per_view.txt
