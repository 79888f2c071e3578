#!/bin/bash
# round 3: the randomised parity test under 24 more seeds (stress run; the committed test runs one seed)
set -e
mkdir -p gpurun_out/r03zr
for seed in $(seq 1 24); do
  VR_TEST_SEED=$seed python -m pytest tests/test_gpu_random.py -x -q -m gpu > gpurun_out/r03zr/seed_$seed.log 2>&1 || { echo "seed $seed FAILED"; tail -30 gpurun_out/r03zr/seed_$seed.log; exit 1; }
  echo "seed $seed: $(tail -1 gpurun_out/r03zr/seed_$seed.log)"
done
