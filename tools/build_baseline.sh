#!/bin/bash
# Build the library of a given commit (default HEAD) into build_exp/lib_base.so for same-box A/B runs
# (PARROT_HIP_LIB=build_exp/lib_base.so python bench.py ...): GPU boxes differ by a few percent.
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/parrot_tts_amd/csrc" "$TMP/include" "$ROOT/build_exp"
for f in $(git -C "$ROOT" ls-tree --name-only "$REV" parrot_tts_amd/csrc/); do git -C "$ROOT" show "$REV:$f" > "$TMP/$f"; done
git -C "$ROOT" show "$REV:include/parrot_hip.h" > "$TMP/include/parrot_hip.h"
(cd "$TMP/parrot_tts_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o "$ROOT/build_exp/lib_base.so" parrot_hip.hip)
rm -rf "$TMP"
echo "$ROOT/build_exp/lib_base.so"
