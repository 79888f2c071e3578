#!/bin/bash
# round 3: are the L2 misses of the run-brick views set conflicts of power-of-two brick strides?  1024^3 (128 bricks per axis) against 1016^3 (127)
# and 1032^3 (129): kernel times and TCC hit / miss / fabric requests per view
set -e
mkdir -p gpurun_out/r03zl
for n in 1024 1016 1032; do
  python scripts/perf_probe.py --reps 4 --volume $n > gpurun_out/r03zl/t_$n.json
  bash scripts/gpu_pmc.sh gpurun_out/r03zl/pmc_$n tcc --volume $n --views 0,1,2,3,4,5,6,7
  python scripts/pmc_per_view.py gpurun_out/r03zl/pmc_$n 6 raymarch 2 > gpurun_out/r03zl/pmc_$n.txt
done
