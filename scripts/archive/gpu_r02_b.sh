#!/bin/bash
# round 2, call B: exact saturation shortcut — parity (whole -m gpu tier on the in-tree build) and A/B timing
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r02b; mkdir -p $OUT
BV=$GRAFT_REPO_ROOT/build_variants
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
for V in sat nosat; do
  echo "== $V lit";     VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py || exit 1
  echo "== $V nearest"; VR_HIP_LIB=$BV/libvr_hip_$V.so timeout -k 10 300 python scripts/perf_probe.py --sampling nearest || exit 1
done
echo "== sat unlit"; VR_HIP_LIB=$BV/libvr_hip_sat.so timeout -k 10 300 python scripts/perf_probe.py --light 0 || exit 1
echo "== sat default"; VR_HIP_LIB=$BV/libvr_hip_sat.so timeout -k 10 300 python scripts/perf_probe.py --mode default || exit 1
echo "== sat noise volume lit"; VR_HIP_LIB=$BV/libvr_hip_sat.so timeout -k 10 300 python scripts/perf_probe.py --kind noise || exit 1
echo "== nosat noise volume lit"; VR_HIP_LIB=$BV/libvr_hip_nosat.so timeout -k 10 300 python scripts/perf_probe.py --kind noise || exit 1
