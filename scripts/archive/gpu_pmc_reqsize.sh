#!/bin/bash
# size mix of the L2 -> fabric read requests of bench.py's kernels (calibration of FETCH_SIZE for the gather pattern)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/reqsize; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/a -- python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-extras > $OUT/a.json 2> $OUT/a.err || { tail -5 $OUT/a.err; exit 1; }
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_READ_sum TCC_READ_SECTORS_sum --kernel-trace --output-format csv -d $OUT/b -- python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-extras > $OUT/b.json 2> $OUT/b.err || { tail -5 $OUT/b.err; exit 1; }
for k in raymarch minmax_kernel generate brickify; do echo "== $k"; python scripts/pmc_summary.py $OUT $k; done
