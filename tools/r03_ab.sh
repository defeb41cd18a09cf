#!/bin/bash
# A/B of env-selected variants on one box:  VARS="PARROT_RB_DUAL=0 PARROT_RB_DUAL=1 ..." tools/r03_ab.sh tag
TAG=${1:-ab}; ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
for v in $VARS; do
  for rep in 1 ${REPS:-}; do
  env $v timeout 600 python bench.py --no-cpu-baseline --no-alt --steps ${STEPS:-10} > $OUT/${TAG}_$v.json 2> $OUT/${TAG}_$v.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/${TAG}_$v.json"))
    print("$v", "ms_per_step %.3f" % d["ms_per_step"], " | ".join("%s %.3f" % (r["kernel"].split("<")[0][-22:]+r["kernel"][-9:], r["ms_per_step"]) for r in d["roofline"]["all_conv_kernels"][:7]))
except Exception as e:
    print("$v", "FAILED", e, open("$OUT/${TAG}_$v.err").read()[-500:])
PY
  done
done
