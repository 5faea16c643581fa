#!/bin/bash
# Builds the library of another commit (default HEAD) into tools/lab/old_lib/libbasic_dsp_hip_B.so -- the baseline the
# A/B scripts (tools/ab_*.sh) time against the tree's library on the same box:   tools/build_baseline.sh [commit]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" | tar -x -C "$TMP"
make -C "$TMP/basic_dsp_amd/csrc" -j"$(nproc)" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
mkdir -p "$ROOT/tools/lab/old_lib"
cp "$TMP/basic_dsp_amd/lib/libbasic_dsp_hip.so" "$ROOT/tools/lab/old_lib/libbasic_dsp_hip_B.so"
rm -rf "$TMP"
echo "baseline of $(git -C "$ROOT" rev-parse --short "$REV") -> tools/lab/old_lib/libbasic_dsp_hip_B.so"
