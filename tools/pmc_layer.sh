#!/bin/bash
# Per-kernel stall / issue counters for one conv layer of the microbenchmark (separate rocprofv3 --pmc passes).
#   tools/pmc_layer.sh <layer> <precision> [batch]      -> gpurun_out/pmc_<layer>_p<precision>/*.csv
set -u
LAYER=${1:-mrf0_k3}; PREC=${2:-1}; BATCH=${3:-64}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_${LAYER}_p${PREC}
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
while read -r CTRS; do
  [ -z "$CTRS" ] && continue
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT/p$i" -o r -- \
      python "$ROOT/tools/microbench_conv.py" --precision "$PREC" --layers "$LAYER" --tiles -1 --batch "$BATCH" --iters 10 > "$OUT/p$i.log" 2>&1
  DB=$(find "$OUT/p$i" -name '*.db' | head -1)
  [ -n "$DB" ] && python "$ROOT/tools/rocpd_summary.py" "$DB" --csv "$OUT/p$i.csv" > /dev/null 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum
LIST
python - "$OUT" <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/p*.csv")):
    for r in csv.DictReader(open(f)):
        if "conv_" in r["kernel"] or "resblock" in r["kernel"]:
            print(f.split("/")[-1], r["kernel"], r["calls"], r["avg_us"], {k[:-9]: float(v) for k, v in r.items() if k.endswith("_per_call") and v})
PY
