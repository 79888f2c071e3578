#!/bin/bash
# round 4, call j: column test under the bounds-checked build first, then the product: parity subset, timings.  Stops at the first failure.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_j; mkdir -p $O
VR_HIP_LIB=$PWD/build_variants/libvr_hip_bounds.so timeout -k 10 150 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column" > $O/tests_bc.log 2>&1 || { grep -E "bounds check|fault|Abort|assert|Error" $O/tests_bc.log | head; exit 1; }
echo "bounds-checked build: $(tail -1 $O/tests_bc.log)"
bash scripts/gpu_r04_g.sh "$@"
