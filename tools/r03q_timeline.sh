#!/bin/bash
# launch timelines of the 16x16x32 layer kernel for profiles/ (needs build_exp/libparrot_s16trace.so: tools/build_exp.sh s16trace -DS16_TRACE)
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export PARROT_HIP_LIB=build_exp/libparrot_s16trace.so
( for l in mrf0_k11d5 mrf0_k7d3 mrf1_k11d5 mrf2_k11d5 ffn1_k9; do python tools/s16_launch_timeline.py --layer $l --batch 64; done
  for v in "PARROT_S16_N160=0" "PARROT_S16_PRIO=1" "PARROT_S16_PRIO=2"; do for l in mrf0_k11d5 mrf1_k11d5; do echo "{\"env\": \"$v\"}"; env $v python tools/s16_launch_timeline.py --layer $l --batch 64; done; done ) 2>&1 | grep -v amdgpu.ids > $OUT/r03q_s16_launch_timeline.jsonl
