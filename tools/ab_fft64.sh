#!/bin/bash
# A/B of the f64 transforms between library A (tools/lab/old_lib) and the tree's on the same box
for rep in 1 2; do
  for lib in tools/lab/old_lib/libbasic_dsp_hip_B.so ""; do
    echo "== lib: ${lib:-tree}"
    for args in "--points 4194304 --elem 1" "--points 2097152 --elem 1" "--points 1048576 --elem 1 --batch 16" "--points 4194304 --elem 1 --batch 4"; do
      BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what fft --iters 200 $args 2>&1 | grep -v amdgpu.ids
    done
    BDSP_HIP_LIBRARY=$lib python3 tools/bench_configs.py 2>/dev/null | grep "C4a"
  done
done
