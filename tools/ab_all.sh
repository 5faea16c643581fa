#!/bin/bash
# A/B of library A (tools/lab/old_lib/libbasic_dsp_hip_B.so) against the tree's over the transforms the round-3 FFT
# changes touch (FMA-form radix-16/8 stages everywhere, folded inter-pass twiddles, split f64 exchange)
for rep in 1 2; do
  for lib in tools/lab/old_lib/libbasic_dsp_hip_B.so ""; do
    echo "== lib: ${lib:-tree}"
    for args in "--points 16777216" "--points 1048576 --batch 64" "--points 1048576" "--points 4096 --batch 4096" "--points 8192 --batch 2048" "--points 65536 --batch 256" "--points 4194304 --elem 1" "--points 1048576 --elem 1 --batch 16" "--points 16777216 --elem 1"; do
      BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what fft --iters 200 $args 2>&1 | grep -v amdgpu.ids
    done
    BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what convfft --iters 400 2>&1 | grep -v amdgpu.ids
    BDSP_HIP_LIBRARY=$lib python3 tools/bench_configs.py 2>/dev/null | grep "C4a\|C2 \|C5"
  done
done
