cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 1500 python3 -m pytest tests -m gpu -q --timeout 600 > gpurun_out/r02c/pytest_gpu.txt 2>&1
tail -15 gpurun_out/r02c/pytest_gpu.txt
python3 tools/kbench.py --what conv,tapsconv,fft --iters 200 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02c/kbench.txt
python3 bench.py --no-cpu-baseline 2>/dev/null | tee gpurun_out/r02c/bench.json
