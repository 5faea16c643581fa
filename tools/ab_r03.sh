#!/bin/bash
# A/B of two builds of the library on the same box: baseline library (tools/build_baseline.sh -> tools/lab/old_lib) against the tree's.
# usage: tools/ab_r03.sh   (on the GPU box)
OLD=tools/lab/old_lib/libbasic_dsp_hip_B.so
for rep in 1 2; do
  for lib in "$OLD" ""; do
    echo "== lib: ${lib:-tree}"
    BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what conv,tapsconv,convfft --iters 400
    BDSP_HIP_LIBRARY=$lib python3 tools/kbench.py --what conv --elem 1 --iters 200
  done
done
BDSP_HIP_LIBRARY=$OLD python3 tools/realconv_bench.py
python3 tools/realconv_bench.py
