#!/bin/bash
# round 4, call d: the whole -m gpu tier (incl. C5 whole frames, race test, bounds build), smoke, then bench.py with every extra leg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_d; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -v 2>&1 | tee $O/tests.log | grep -E "PASSED|FAILED|ERROR|passed|failed|rror" | tail -70
grep -q "failed" $O/tests.log && exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04_d/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms'], 'per view', d['roofline'].get('per_view_kernel_ms'))
ex=d.get('extras',{})
for k in ('nooptims_nearest','default_trilinear','default_nearest','ertonly_trilinear','ertonly_nearest'):
    if k in ex: print(k, ex[k]['kernel_ms'])
fv=ex.get('first_visit',{})
for k,v in fv.items():
    if isinstance(v,dict): print('first_visit',k,{kk:vv for kk,vv in v.items() if 'per_view' not in kk})
print('set_volume', json.dumps(d.get('set_volume'))[:1500])
print('configs', {k:(v.get('nooptims',{}).get('kernel_ms'), v.get('default',{}).get('kernel_ms'), v.get('setup_s')) for k,v in ex.get('configs',{}).items() if isinstance(v,dict)})
print('host_buffer_ms', d.get('host_buffer_ms'))
PY
