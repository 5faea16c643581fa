#!/bin/bash
# GPU box: the two-trip FFT lab.  usage: run_fft2trip.sh "<bin> <args>" ...   (each entry one run)
cd $GRAFT_REPO_ROOT/tools/lab
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/lab
OUT=$GRAFT_REPO_ROOT/gpurun_out/lab/fft2trip_$(date +%H%M%S).txt
{
for run in "$@"; do
  echo "=== $run"
  timeout 120 ./$run
done
} 2>&1 | tee $OUT
