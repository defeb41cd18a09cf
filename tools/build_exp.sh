#!/bin/bash
# Ablation builds of the library (same sources + -DEXP_* switches) into build_exp/ (git-ignored, shipped by gpurun):
#   tools/build_exp.sh NO_STORE NO_BARRIER "NO_STORE -DEXP_NO_BARRIER" ...   ->  build_exp/libparrot_<tag>.so
# select with PARROT_HIP_LIB=build_exp/libparrot_<tag>.so.  Timing experiments only: results are wrong.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build_exp"
for spec in "$@"; do
  tag=$(echo "$spec" | tr -d ' -' | sed 's/DEXP_/_/g')
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DEXP_$spec -o "$ROOT/build_exp/libparrot_$tag.so" "$ROOT/parrot_tts_amd/csrc/parrot_hip.hip" && echo built $tag ) &
done
wait
