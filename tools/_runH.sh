python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python3 tools/interp_frac_kernel.py 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/interp_frac_r05.txt; cat gpurun_out/interp_frac_r05.txt
