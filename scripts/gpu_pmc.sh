#!/bin/bash
# PMC counter passes over the ray-march kernel. Usage: gpu_pmc.sh <outdir> <passes: comma list> [probe args...]
OUT=${1:-gpurun_out/pmc}; PASSES=${2:-sq1,sq2,tcp1,tcc,fetch}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
PROBE_ARGS="${@:---views 0}"
pass() { # name counters...
  local name=$1; shift
  case ",$PASSES," in *",$name,"*) ;; *) return 0;; esac
  timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python scripts/perf_probe.py --reps 2 $PROBE_ARGS > $OUT/$name.log 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -ne 0 ]; then tail -3 $OUT/$name.log; exit 1; fi   # never start another GPU step after a killed one
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
pass sq2 SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
pass ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
exit 0
