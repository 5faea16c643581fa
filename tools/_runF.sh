python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python3 tools/interp_frac_kernel.py 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/interp_frac_r05.txt
SECTIONS="K" bash tools/plan_matrix.sh > gpurun_out/plan_matrix.log 2>&1
cp gpurun_out/plan_matrix.txt gpurun_out/plan_matrix_K.txt
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r05a.json 2> gpurun_out/bench_r05a.err; tail -c 300 gpurun_out/bench_r05a.err
