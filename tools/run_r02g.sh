cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests -m gpu -q -k "fft or FFT or c2 or c4 or 16m or interpolate" --timeout 600 2>&1 | tail -3
python3 tools/kbench.py --what fft --points 1048576 --batch 64 --iters 50 2>&1 | grep -v amdgpu
python3 tools/kbench.py --what fft --points 1048576 --batch 1 --iters 200 2>&1 | grep -v amdgpu
python3 tools/kbench.py --what fft --iters 100 2>&1 | grep -v amdgpu
python3 tools/kbench.py --what fft --points 4194304 --elem 1 --iters 100 2>&1 | grep -v amdgpu
BDSP_FFT_NO_PERSIST=1 python3 tools/kbench.py --what fft --iters 100 2>&1 | grep -v amdgpu
BDSP_FFT_NO_PERSIST=1 python3 tools/kbench.py --what fft --points 1048576 --batch 64 --iters 50 2>&1 | grep -v amdgpu
