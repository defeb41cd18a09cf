#!/bin/bash
# Experiment builds of the library (same sources + -D switches) into build_exp/ (git-ignored, shipped by gpurun):
#   tools/build_exp.sh trace -DRBD_TRACE          ->  build_exp/libparrot_trace.so
# select with PARROT_HIP_LIB=build_exp/libparrot_<tag>.so.  Timing / tracing experiments only.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT" && mkdir -p build_exp && python -m parrot_tts_amd.build --exp "$@"
