#!/bin/bash
# Collect the per-round evidence on the GPU box: kernel-trace stats and separate PMC passes of the SAME bench command.
#   tools/profile_round.sh r01d      -> gpurun_out/<tag>_*.csv (copy into profiles/ afterwards), gpurun_out/<tag>_bench_n1.json
set -u
TAG=${1:-rXX}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
# which box these counters come from (bench.py prints it beside roofline.traffic)
{ rocminfo 2>/dev/null | grep -m1 -i "Uuid:.*GPU-" | sed 's/^ *//'; rocminfo 2>/dev/null | grep -m1 -i "Marketing Name:.*MI3" | sed 's/^ *//'; rocm-smi --showuniqueid 2>/dev/null | grep -i "unique id:" | head -1; } | tr '\n' ' ' | sed 's/  */ /g' > "$OUT/${TAG}_box.txt"
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt"
cd /tmp
run() {  # name, rocprofv3 args...
  local name=$1; shift
  rm -rf "$OUT/prof_$name"
  timeout 600 rocprofv3 "$@" -d "$OUT/prof_$name" -o r -- $CMD > "$OUT/prof_$name.log" 2>&1
  local db=$(find "$OUT/prof_$name" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/rocpd_summary.py" "$db" --csv "$OUT/${TAG}_bench_$name.csv"
  [ "$name" = stats ] && [ -n "$db" ] && python "$ROOT/tools/rocpd_summary.py" "$db" --by-grid --csv "$OUT/${TAG}_bench_stats_by_grid.csv"
  rm -rf "$OUT/prof_$name"
}
run stats --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT
cd "$ROOT"
timeout 400 python bench.py > "$OUT/${TAG}_bench_n1.json" 2> "$OUT/${TAG}_bench_n1.err"
tail -c 600 "$OUT/${TAG}_bench_n1.json"
