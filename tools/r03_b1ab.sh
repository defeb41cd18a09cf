#!/bin/bash
# A/B of env-selected variants at small batches:  VARS="A=0 A=1" BATCHES="1 4 16" tools/r03_b1ab.sh tag
TAG=${1:-b1ab}; ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
for b in ${BATCHES:-1}; do for v in $VARS; do for rep in 1 ${REPS:-}; do
  env $v timeout 600 python bench.py --batch $b --no-cpu-baseline --no-alt --steps ${STEPS:-30} --warmup 5 > $OUT/${TAG}_${b}_$v.json 2> $OUT/${TAG}_${b}_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_${b}_$v.json")); print("B=$b $v ms_per_step %.3f" % d["ms_per_step"])
except Exception as e:
    print("$v", "FAILED", e, open("$OUT/${TAG}_${b}_$v.err").read()[-500:])
PY
done; done; done
