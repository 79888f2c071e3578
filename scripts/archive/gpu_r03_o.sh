#!/bin/bash
# after the per-slot streams: parity tier (multi-device tests), bench under torch.distributed.run with one rank and TWO slot streams, the full bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_driver.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -1 $O/tests.log
VR_BENCH_TWO_STREAMS=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_dist1.json 2> $O/bench_dist1.err || { tail -30 $O/bench_dist1.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_dist1.json')); print('dist1 two streams:', d['ms_per_step'], d['frame_check'], d['config']['partition'], d['roofline']['kernel_ms'])"
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --mode default > $O/bench_dist1d.json 2> $O/bench_dist1d.err || { tail -30 $O/bench_dist1d.err; exit 1; }
python -c "import json; d=json.load(open('$O/bench_dist1d.json')); print('dist1 default:', d['ms_per_step'], d['frame_check'])"
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json'))
print(d['value'], d['ms_per_step'], d['frame_check'])
for m in ('nooptims','default'):
  print(m, {k:(v['predicted_efficiency'], v['predicted_efficiency_pipelined'], v['pipelined_ms_per_frame']) for k,v in d['scale_model'][m].items() if k.startswith('n')})
print({k:(v['sync_ms_per_frame'], v['pipelined_ms_per_frame']) for k,v in d['extras']['multi_overhead'].items() if k.startswith('n')})
"
