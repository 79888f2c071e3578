#!/bin/bash
# round 3: both run copies per tile (kLayoutRunDual): parity, then the oblique views with the measured choice against one copy
set -e
mkdir -p gpurun_out/r03zf
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "layouts_agree or scheduling" > gpurun_out/r03zf/pytest.log 2>&1 || { tail -30 gpurun_out/r03zf/pytest.log; exit 1; }
tail -2 gpurun_out/r03zf/pytest.log
python scripts/perf_probe.py --reps 8 --views 1,5 --each > gpurun_out/r03zf/auto.json
python scripts/perf_probe.py --reps 8 --views 1,5 --plane 3 > gpurun_out/r03zf/runz.json
python scripts/perf_probe.py --reps 8 --views 1,5 --plane 7 > gpurun_out/r03zf/alternating.json
python scripts/perf_probe.py --reps 8 --views 0,1,2,3,4,5,6,7 --plane 6 > gpurun_out/r03zf/dual_all.json
python scripts/perf_probe.py --reps 8 > gpurun_out/r03zf/auto_all.json
