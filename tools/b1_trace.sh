#!/bin/bash
# kernel trace of the single-utterance pipeline (BASELINE configs[0] shape):  tools/b1_trace.sh tag  -> gpurun_out/<tag>_b1_by_grid.csv
TAG=${1:-b1}; ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/pb1
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pb1 -o r -- python $ROOT/bench.py --batch 1 --steps 20 --warmup 3 --no-cpu-baseline --no-alt > $OUT/${TAG}_b1.json 2> $OUT/${TAG}_b1.err
python $ROOT/tools/rocpd_summary.py $(find /tmp/pb1 -name '*.db' | head -1) --by-grid --csv $OUT/${TAG}_b1_by_grid.csv
python - <<PY
import json,csv
d=json.load(open("$OUT/${TAG}_b1.json")); print("ms_per_step", d["ms_per_step"])
rows=list(csv.DictReader(open("$OUT/${TAG}_b1_by_grid.csv")))
tot=sum(float(r["total_ms"]) for r in rows)/23
print("kernel ms per step", tot)
for r in rows[:28]: print(r["kernel"][:60], r["wg_x"], r["wg_y"], round(int(r["calls"])/23,1), r["avg_us"], round(float(r["total_ms"])/23,3))
PY
