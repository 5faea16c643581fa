SECTIONS="J" bash tools/plan_matrix.sh > gpurun_out/plan_matrix.log 2>&1
cp gpurun_out/plan_matrix.txt gpurun_out/plan_matrix_J.txt
export BDSP_HIP_LIBRARY=basic_dsp_amd/lib/libbasic_dsp_hip_lab_ntl.so
for rep in 1 2; do timeout 300 python3 tools/conv_probe.py f64 2>&1 | grep -v amdgpu.ids | grep "default\|streamed stores  \|headline" | sed 's/^/NT-LOADS /'; done > gpurun_out/conv_ntl.txt 2>&1
tail -3 gpurun_out/conv_ntl.txt
