#!/bin/bash
# round 4, call g: column march (lean window step): parity, then views 0 / 2 / 3 lit and unlit, then all eight views.  Stops at the first step that fails or times out.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_g; mkdir -p $O
timeout -k 10 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column" > $O/tests0.log 2>&1 || { tail -15 $O/tests0.log; exit 1; }
tail -2 $O/tests0.log
timeout -k 10 300 python -m pytest tests/test_gpu_copies.py tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q -k "copies or layouts_agree or volume_info or random or nearest_bit_exact" > $O/tests.log 2>&1 || { tail -25 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for light in 0.6 0.0; do
  echo "== product light $light" | tee -a $O/probe.log
  timeout -k 10 60 python scripts/perf_probe.py --mode nooptims --views 0,2,3 --light $light --reps 6 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
  cat $O/line.json >> $O/probe.log; cut -c100-260 $O/line.json
done
for lib in "$@"; do
  export VR_HIP_LIB=$PWD/build_variants/libvr_hip_$lib.so
  echo "== $lib light 0.6" | tee -a $O/probe.log
  timeout -k 10 60 python scripts/perf_probe.py --mode nooptims --views 0,2,3 --light 0.6 --reps 6 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
  cat $O/line.json >> $O/probe.log; cut -c100-260 $O/line.json
done
unset VR_HIP_LIB
timeout -k 10 100 python scripts/perf_probe.py --mode nooptims --reps 6 2>> $O/probe.err > $O/line.json || { tail -5 $O/probe.err; exit 1; }
cat $O/line.json >> $O/probe.log; cut -c100-330 $O/line.json
