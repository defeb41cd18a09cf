#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export PARROT_HIP_LIB=$ROOT/build_exp/libparrot_trace.so PARROT_RB_DUAL=1
for sel in 3203 3211 1603 1611 6403; do
  echo "== sel $sel"; PARROT_RBD_TRACE_SEL=$sel timeout 300 python tools/rbd_trace.py 2>&1 | tail -20
done > $OUT/r03c_trace.log 2>&1
tail -5 $OUT/r03c_trace.log
