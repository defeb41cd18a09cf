#!/bin/bash
# Deeper SQ counter passes of the bench command, per (kernel, grid) = per layer shape:
#   tools/profile_sq_deep.sh r03a   ->  gpurun_out/<tag>_sqdeep_{a,b,c}.csv   (copy into profiles/ afterwards)
# Counters only (--kernel-trace + --pmc; never combined with the other trace domains).
set -u
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt ${BENCH_ARGS:-}"
cd /tmp
run() {
  local name=$1; shift
  rm -rf "$OUT/prof_$name"
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/prof_$name" -o r -- $CMD > "$OUT/prof_$name.log" 2>&1
  local db=$(find "$OUT/prof_$name" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/rocpd_summary.py" "$db" --by-grid --csv "$OUT/${TAG}_sqdeep_$name.csv"
  rm -rf "$OUT/prof_$name"
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES
run b SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD
run c SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
