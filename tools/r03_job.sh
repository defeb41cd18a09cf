#!/bin/bash
# generic round-3 GPU job: tests then a bench line:  tools/r03_job.sh <tag> [pytest args...]
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"; cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -x -q "$@" > "$OUT/${TAG}_pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/${TAG}_pytest.log"
tail -5 "$OUT/${TAG}_pytest.log"
timeout 900 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
python - <<PY
import json
d=json.load(open("$OUT/${TAG}_bench.json"))
print("ms_per_step", d["ms_per_step"], "value", d["value"])
for r in d["roofline"]["all_conv_kernels"]: print(r["kernel"], round(r["ms_per_step"],3), round(r["tflops"],1), r["launches_per_step"])
for k in ("pipelined_steps","vocoder_only_b32","single_utterance_b1"):
    if k in d: print(k, d[k]["ms_per_step"])
PY
