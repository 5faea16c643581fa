cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
(cd tools/ubench && timeout 300 ./mem_interference) > gpurun_out/r02a/mem_interference.txt 2>&1
(cd tools/ubench && timeout 120 ./cluster_pass) > gpurun_out/r02a/cluster_pass.txt 2>&1
timeout 300 python3 bench.py --no-cpu-baseline > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
cat gpurun_out/r02a/mem_interference.txt gpurun_out/r02a/cluster_pass.txt gpurun_out/r02a/bench.json
