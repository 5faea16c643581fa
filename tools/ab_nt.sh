#!/bin/bash
# Non-temporal stores A/B: the tree's library against builds with -DBDSP_FFT_NT / -DBDSP_CONV_NT (tools/lab/old_lib)
for rep in 1 2; do
  for lib in "" tools/lab/old_lib/libbasic_dsp_hip_fftnt.so tools/lab/old_lib/libbasic_dsp_hip_convnt.so; do
    echo "== lib: ${lib:-tree}"
    for args in "--points 16777216 --elem 1" "--points 16777216" "--points 1048576 --batch 64" "--points 4194304 --elem 1 --batch 4" "--points 33554432" "--points 67108864"; do
      BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what fft --iters 100 $args 2>&1 | grep -v amdgpu.ids
    done
    BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what conv,convfft --iters 300 2>&1 | grep -v amdgpu.ids
    BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what conv --points 1048576 --batch 64 --iters 50 2>&1 | grep -v amdgpu.ids
    BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what conv --points 67108864 --iters 50 2>&1 | grep -v amdgpu.ids
    BDSP_HIP_LIBRARY=$lib python3 tools/bench_configs.py 2>/dev/null | grep "C2x64\|C5\|C3 in f64"
  done
done
