#!/bin/bash
# A/B of interpolatef between the baseline library (tools/build_baseline.sh) and the tree's on the same box
for rep in 1 2; do
  for lib in tools/lab/old_lib/libbasic_dsp_hip_B.so ""; do
    echo "== lib: ${lib:-tree}"
    BDSP_HIP_LIBRARY=$lib python3 tools/interp_bench.py 2>&1 | grep -v amdgpu.ids
  done
done
