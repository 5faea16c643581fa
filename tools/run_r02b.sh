cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
{
python3 tools/kbench.py --what conv,tapsconv --iters 200
python3 tools/kbench.py --what conv,tapsconv --iters 200 --zero
python3 tools/kbench.py --what fft --iters 200
python3 tools/kbench.py --what fft --iters 200 --zero
} > gpurun_out/r02b/kbench.txt 2>&1
cat gpurun_out/r02b/kbench.txt
