#!/bin/bash
# GPU box: non-temporal LOADS in the FFT pass kernels (LAB build with -DBDSP_FFT_NTLOAD) against the plain LAB build
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in basic_dsp_amd/lib/libbasic_dsp_hip_lab.so basic_dsp_amd/lib/libbasic_dsp_hip_lab_ntload.so; do
    echo "== lib: $lib"
    for args in "--points 16777216" "--points 1048576 --batch 64" "--points 4194304 --elem 1" "--points 33554432"; do
      BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what fft --iters 100 $args 2>&1 | grep -v amdgpu.ids
    done
    BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what convfft --iters 300 2>&1 | grep -v amdgpu.ids
  done
done
