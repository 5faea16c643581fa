#!/bin/bash
# A/B of interpolatef between the round-2 library and the tree's on the same box
for rep in 1 2; do
  for lib in tools/lab/old_lib/libbasic_dsp_hip_r02.so ""; do
    echo "== lib: ${lib:-tree}"
    BDSP_HIP_LIBRARY=$lib python3 tools/interp_bench.py 2>&1 | grep -v amdgpu.ids
  done
done
