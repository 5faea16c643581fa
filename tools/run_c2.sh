cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
{
for b in 64 32 16 8 4 1; do python3 tools/kbench.py --what fft --points 1048576 --batch $b --iters 100 2>&1 | grep -v amdgpu; done
for b in 64 16; do python3 tools/kbench.py --what conv --points 1048576 --batch $b --iters 100 2>&1 | grep -v amdgpu; done
} | tee gpurun_out/r02d/c2_batches.txt
