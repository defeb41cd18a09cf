#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export PARROT_RB_DUAL=1
for v in ${VARIANTS:-trace NO_A NO_B NO_MFMA NO_WP}; do
 for sel in ${SELS:-3203}; do
  echo "== variant $v sel $sel"; PARROT_HIP_LIB=$ROOT/build_exp/libparrot_$v.so PARROT_RBD_TRACE_SEL=$sel timeout 300 python tools/rbd_trace.py 2>&1 | grep -A1 "^wave [04] "
 done
done > $OUT/${TAG:-r03d}_trace.log 2>&1
cat $OUT/${TAG:-r03d}_trace.log
