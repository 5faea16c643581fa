cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
(cd tools/ubench && ./permlane_exchange) 2>&1 | tee gpurun_out/r02f/permlane_exchange.txt
(cd tools/lab && ./conv_lab 16777216 1024) 2>&1 | grep "^base  \|L3\|st tw0 3 rounds 9/8\|v2 st tw0 3 wg" | cut -c1-120 | tee gpurun_out/r02f/lab_l3.txt
timeout 900 python3 -m pytest tests -m gpu -q -k "convolve or conv" --timeout 600 2>&1 | tail -3
