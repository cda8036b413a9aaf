#!/bin/bash
# PMC passes over a short bench run (one 16-spp step at 1080p, 8 bounces). Each pass is its own rocprofv3 run.
# usage: tools/profile_pmc.sh <outdir>
# RT_PMC_ARGS='--workload spheres10m' adds bench.py arguments; RT_PMC_PASSES='fetch write' runs only those passes
set -u
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
export TMPDIR=/tmp
export RT_TUNE_ONE_STREAM=1 # counter collection serialises kernels: profile the one-stream launch geometry (full traversal grid)
run() {
  name=$1; shift
  case " ${RT_PMC_PASSES:-sq1 sq2 tcc1 ta fetch write} " in *" $name "*) ;; *) return 0;; esac
  timeout -s KILL 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-side ${RT_PMC_ARGS:-} > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_INSTS_FLAT &&
run tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum &&
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
