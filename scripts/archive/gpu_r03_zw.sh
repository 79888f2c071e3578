#!/bin/bash
# round 3: two ESL probes per turn where the first leaps by 0 (VR_ESL_PROBE_PAIRS): parity of everything that leaps, then the default mode per view,
# several processes for the view whose time depends on what the order recordings caught
set -e
mkdir -p gpurun_out/r03zw
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r03zw/pytest.log 2>&1 || { tail -30 gpurun_out/r03zw/pytest.log; exit 1; }
tail -2 gpurun_out/r03zw/pytest.log
for i in 1 2 3 4 5 6; do python scripts/perf_probe.py --mode default --views 5,3 --reps 12 | python -c "import json,sys; d=json.load(sys.stdin); print('pairs', d['kernel_ms_per_view'])"; done
python scripts/perf_probe.py --mode default --reps 8 | python -c "import json,sys; d=json.load(sys.stdin); print('all', d['mean_ms'], d['kernel_ms_per_view'])"
python scripts/perf_probe.py --mode default --sampling nearest --reps 8 | python -c "import json,sys; d=json.load(sys.stdin); print('all nearest', d['mean_ms'], d['kernel_ms_per_view'])"
python scripts/perf_probe.py --mode default --sched 0 --reps 8 | python -c "import json,sys; d=json.load(sys.stdin); print('all sched0', d['mean_ms'], d['kernel_ms_per_view'])"
python scripts/perf_probe.py --reps 6 | python -c "import json,sys; d=json.load(sys.stdin); print('full march', d['mean_ms'], d['kernel_ms_per_view'])"
