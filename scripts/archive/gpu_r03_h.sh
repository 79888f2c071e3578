#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['kernel_ms_max'], d['roofline']['per_view_kernel_ms'])
print({k:(v.get('kernel_ms'), v.get('kernel_ms_max')) for k,v in d['extras'].items() if isinstance(v,dict) and 'kernel_ms' in v})
print('host_buffer', d['extras']['host_buffer'])
for k,v in d['extras']['configs'].items(): print(k, {m:(v[m]['kernel_ms'], v[m]['kernel_ms_max']) for m in ('nooptims','default') if m in v}, v.get('skipped'), v.get('copies_built'), v.get('setup_s'))
print('linear', d['extras']['linear_layout']['kernel_ms'], d['extras']['linear_layout']['per_view_kernel_ms'])
print('multi', d['extras']['multi_overhead'])
print('set_volume', d['set_volume'])
for m in ('nooptims','default'):
  print(m, {k:(v['max_over_mean'], v['predicted_efficiency'], v['predicted_ms_per_frame']) for k,v in d['scale_model'][m].items() if k.startswith('n')})
print(d['cpu_baseline']['value'], d['vs_cpu_baseline'])
"
