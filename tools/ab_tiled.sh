#!/bin/bash
# GPU box: A/B of the tiled two-pass intermediate (LAB library; BDSP_FFT_TILED=1 = the tiled intermediate)
export BDSP_HIP_LIBRARY=$GRAFT_REPO_ROOT/basic_dsp_amd/lib/libbasic_dsp_hip_lab.so
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "" 1; do
    if [ -n "$v" ]; then unset BDSP_FFT_TILED; echo "== natural-order intermediate"; else export BDSP_FFT_TILED=1; echo "== tiled intermediate"; fi
    python tools/c2_half512_check.py 2>&1 | grep -v amdgpu.ids
    python tools/bench_configs.py 2>/dev/null | grep -E "C2x64|C4a|C5/GPU|\"C2 complex"
  done
done
