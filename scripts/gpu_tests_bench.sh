#!/bin/bash
# The GPU tier, smoke() and the default bench line in one call; stops at the first failing step.  usage: gpu_tests_bench.sh [outdir]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/tb}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -25 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -5 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "kernel", d.get("kernel_ms"), "host_buffer", d.get("host_buffer_ms"))
print("per view", d.get("scale_model", {}).get("nooptims", {}).get("t1_per_view_ms"))
e = d.get("extras", {})
print({k: (v.get("ms_mean") if isinstance(v, dict) else v) for k, v in e.items() if k in ("nooptims_nearest", "default_trilinear", "default_nearest", "ertonly_trilinear", "ertonly_nearest", "host_buffer")})
PY
