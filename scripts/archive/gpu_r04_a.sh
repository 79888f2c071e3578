#!/bin/bash
# round 4, call a: new copy builders (byte-for-byte against the host construction), parity suite, builder timings, C2 whole frames
set -o pipefail
mkdir -p gpurun_out/r04_a
timeout -k 10 600 python -m pytest tests/test_gpu_copies.py -m gpu -x -q > gpurun_out/r04_a/copies.log 2>&1 || { tail -30 gpurun_out/r04_a/copies.log; exit 1; }
tail -3 gpurun_out/r04_a/copies.log
timeout -k 10 300 python scripts/copy_build_probe.py 1024 1 3 > gpurun_out/r04_a/probe_1024_u8.json 2> gpurun_out/r04_a/probe.err || { tail -20 gpurun_out/r04_a/probe.err; exit 1; }
cat gpurun_out/r04_a/probe_1024_u8.json
timeout -k 10 300 python scripts/copy_build_probe.py 1024 2 2 > gpurun_out/r04_a/probe_1024_u16.json 2>> gpurun_out/r04_a/probe.err && cat gpurun_out/r04_a/probe_1024_u16.json
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py -m gpu -x -q > gpurun_out/r04_a/parity.log 2>&1 || { tail -30 gpurun_out/r04_a/parity.log; exit 1; }
tail -3 gpurun_out/r04_a/parity.log
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config2 or config3" > gpurun_out/r04_a/c2c3.log 2>&1 || { tail -30 gpurun_out/r04_a/c2c3.log; exit 1; }
tail -3 gpurun_out/r04_a/c2c3.log
