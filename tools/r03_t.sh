#!/bin/bash
# run selected GPU tests:  tools/r03_t.sh tag "<pytest selection args>"
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p "$OUT"; cd "$ROOT"
timeout 1500 python -m pytest -m gpu -x -q "$@" > "$OUT/${TAG}_pytest.log" 2>&1; echo "rc=$?" >> "$OUT/${TAG}_pytest.log"; tail -25 "$OUT/${TAG}_pytest.log"
