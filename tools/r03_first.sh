#!/bin/bash
# round-3 opening measurement: power-limit probe (+ clock trace), baseline bench with a clock trace, deep SQ counters
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
rocm-smi --showpower --showclocks --showmaxpower > "$OUT/r03a_smi_idle.txt" 2>&1
tools/clock_probe.sh "$OUT/r03a_mfma_power_clk.log" -- build_exp/mfma_power > "$OUT/r03a_mfma_power.jsonl" 2> "$OUT/r03a_mfma_power.err"
tools/clock_probe.sh "$OUT/r03a_bench_clk.log" -- timeout 900 python bench.py > "$OUT/r03a_bench_n1.json" 2> "$OUT/r03a_bench_n1.err"
tools/profile_sq_deep.sh r03a
tail -c 400 "$OUT/r03a_bench_n1.json"
