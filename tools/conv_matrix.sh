#!/bin/bash
# GPU box: the block kernel's round-5 scheduling experiments (tools/conv_probe.py) -> gpurun_out/conv_matrix.txt
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
OUT=gpurun_out/conv_matrix.txt
export BDSP_HIP_LIBRARY=basic_dsp_amd/lib/libbasic_dsp_hip_lab.so
date > $OUT
for rep in 1 2; do
  timeout 300 python3 tools/conv_probe.py f64 2>&1 | grep -v amdgpu.ids | tee -a $OUT
  timeout 300 python3 tools/conv_probe.py c5 2>&1 | grep -v amdgpu.ids | tee -a $OUT
  for g in 3 2; do for prep in 0 1; do
    ( [ $prep = 1 ] && export BDSP_CONV_REAL_PREP=1; [ $g != 3 ] && export BDSP_CONV_GROUPS=$g; timeout 300 python3 tools/conv_probe.py real 2>&1 | grep -v amdgpu.ids | tee -a $OUT )
  done; done
done
# non-temporal LOADS of the block kernel's input (another LAB library): f64 16M, with and without streamed stores
for rep in 1 2; do
  BDSP_HIP_LIBRARY=basic_dsp_amd/lib/libbasic_dsp_hip_lab_ntl.so timeout 300 python3 tools/conv_probe.py f64 2>&1 | grep -v amdgpu.ids | grep "default\|streamed stores  \|headline" | sed 's/^/NT-LOADS /' | tee -a $OUT
done
date >> $OUT
