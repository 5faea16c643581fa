cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02e
timeout 900 python3 -m pytest tests -m gpu -q -k "interp" --timeout 600 2>&1 | tail -3
python3 tools/interp_bench.py 2>&1 | grep -v amdgpu | tail -12
for plan in "" "1024x4,1024x4" "64x64,128x32,128x32" "256x16,4096x4"; do
  echo "plan=[$plan]"; BDSP_FFT_PLAN="$plan" python3 tools/kbench.py --what fft --points 1048576 --batch 64 --iters 50 2>&1 | grep -v amdgpu
done
