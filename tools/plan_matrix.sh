#!/bin/bash
# GPU box: every plan decision of the power-of-two transform re-measured on VALID data, cold and cache-resident
# (tools/plan_probe.py).  Needs tools/build_labs.sh's libraries.  -> gpurun_out/plan_matrix.txt
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
OUT=gpurun_out/plan_matrix.txt
L=basic_dsp_amd/lib/libbasic_dsp_hip_lab
pp() { # pp <lib suffix or ""> <env assignments or -> <tag> args...
  local lib=$L$1.so envs=$2 tag=$3; shift 3
  env BDSP_HIP_LIBRARY=$lib $( [ "$envs" != "-" ] && echo $envs ) timeout 300 python3 tools/plan_probe.py --tag "$tag" "$@" 2>&1 | grep -v "amdgpu.ids" | tee -a $OUT
}
SECTIONS=${SECTIONS:-A B C E F G}   # (D measured the chunked batch schedule, which round 5 removed together with its switches)
date > $OUT
for S in $SECTIONS; do case $S in
A) echo "## A. two against three passes (batch 1, plain complex transform)" | tee -a $OUT
   pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec f32 --bits 20 --plans default,128x32:128x32:64x64
   pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec f32 --bits 21 --plans default,128x32:128x32:128x32,1024x4:2048x4,2048x4:1024x8
   pp "" - product --prec f32 --bits 21 --plans default
   pp "" - product --prec f32 --bits 22 --plans default,256x16:128x32:128x32,1024x8:4096x4,4096x4:1024x8,2048x2:2048x4
   pp "" - product --prec f32 --bits 23 --plans default,4096x4:2048x4,2048x4:4096x4,4096x2:2048x4
   pp "" - product --prec f32 --bits 24 --plans default,4096x4:4096x4
   pp "" - product --prec f64 --bits 20 --plans default,128x32:128x32:64x64
   pp "" - product --prec f64 --bits 21 --plans default,128x32:128x32:128x32,1024x8:2048x4,2048x4:1024x8
   pp "" - product --prec f64 --bits 22 --plans default,256x16:128x32:128x32,2048x2:2048x4,2048x4:2048x2,1024x8:4096x2
   pp "" - product --prec f64 --bits 23 --plans default,4096x2:2048x4,2048x4:4096x2 ;;
B) echo "## B. the last pass of a two-pass plan in place (a -> b -> b) or not (a -> b -> a)" | tee -a $OUT
   for p in f32 f64; do
     pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec $p --bits 14,16,18,20,21,22
     pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec $p --bits 14,16,18,20,21,22
   done ;;
C) echo "## C. f64 tiles: split exchange (two workgroups per CU) or whole complex values (one)" | tee -a $OUT
   for v in "" _nosplit; do
     pp "$v" - "split${v:-_on}" --prec f64 --bits 21,22
     pp "$v" - "split${v:-_on}" --prec f64 --bits 22 --flags 2 --window 4 0.5
     pp "$v" - "split${v:-_on}" --prec f64 --bits 20 --batch 16
   done ;;
D) echo "## D. a large batch in Infinity-Cache-sized chunks (64 x 2^20 f32)" | tee -a $OUT
   for fl in 0 8; do
     pp "" BDSP_FFT_NO_CHUNKS=1 one-piece --prec f32 --bits 20 --batch 64 --flags $fl
     for mb in 64 128 256; do pp "" BDSP_FFT_CHUNK_MB=$mb chunks-of-$((mb/8)) --prec f32 --bits 20 --batch 64 --flags $fl; done
   done
   pp "" BDSP_FFT_NO_CHUNKS=1 one-piece --prec f64 --bits 20 --batch 32
   pp "" BDSP_FFT_CHUNK_MB=128 chunks-of-8 --prec f64 --bits 20 --batch 32 ;;
E) echo "## E. persistent workgroup-resident kernels against the plain launch / two passes" | tee -a $OUT
   for p in f32 f64; do
     pp "" - wg_batch --prec $p --bits 10,11,12 --batch 16384
     pp "" BDSP_FFT_NO_WGBATCH=1 plain-k_fft_wg --prec $p --bits 10,11,12 --batch 16384
     pp "" - wg_batch --prec $p --bits 12 --batch 4096
     pp "" BDSP_FFT_NO_WGBATCH=1 plain-k_fft_wg --prec $p --bits 12 --batch 4096
   done
   for b in 1 256 2048; do
     pp "" - k_fft_wg4 --prec f32 --bits 13 --batch $b
     pp "" BDSP_FFT_NO_WG4=1 two-passes --prec f32 --bits 13 --batch $b
   done ;;
F) echo "## F. config C4a: 4M-point f64 windowed_fft(Hann) + fft_shift" | tee -a $OUT
   pp "" - lab --prec f64 --bits 22 --flags 2 --window 4 0.5 --plans default,256x16:128x32:128x32,1024x8:4096x2,2048x2:2048x4,2048x4:2048x2
   pp _ntload2 - nt-load-first-pass --prec f64 --bits 22 --flags 2 --window 4 0.5 --plans default,256x16:128x32:128x32,1024x8:4096x2 ;;
G) echo "## G. non-temporal stores (every pass) / non-temporal loads (first pass only)" | tee -a $OUT
   for v in "" _nt _ntload2; do
     pp "$v" - "lab$v" --prec f32 --bits 24
     pp "$v" - "lab$v" --prec f32 --bits 20 --batch 64
     pp "$v" - "lab$v" --prec f64 --bits 22
     pp "$v" - "lab$v" --prec f32 --bits 25
   done ;;
H) echo "## H. second look at what run 1 flipped (in-place last pass, f32 2^22 plans, wg_batch in f32), twice" | tee -a $OUT
   for rep in 1 2; do
     for p in f32 f64; do
       pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec $p --bits 15,17,19,20,21,22
       pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec $p --bits 15,17,19,20,21,22
     done
     pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec f32 --bits 22 --plans default,256x16:128x32:128x32,4096x4:1024x8,1024x8:4096x4
     pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec f32 --bits 22 --plans default,4096x4:1024x8,1024x8:4096x4
     pp "" "BDSP_FFT_NO_LAST_INPLACE=1 BDSP_FFT_NO_CHUNKS=1" out-of-place --prec f64 --bits 20 --batch 16
     pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec f64 --bits 20 --batch 16
     pp "" "BDSP_FFT_NO_LAST_INPLACE=1 BDSP_FFT_NO_CHUNKS=1" out-of-place --prec f32 --bits 20 --batch 64
     pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec f32 --bits 20 --batch 64
     pp "" "BDSP_FFT_NO_LAST_INPLACE=1 BDSP_FFT_NO_CHUNKS=1" out-of-place --prec f32 --bits 16 --batch 256
     pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec f32 --bits 16 --batch 256
     for b in 4096 16384; do
       pp "" - wg_batch --prec f32 --bits 10,11,12 --batch $b
       pp "" BDSP_FFT_NO_WGBATCH=1 plain-k_fft_wg --prec f32 --bits 10,11,12 --batch $b
     done
   done ;;
I) echo "## I. config C4a: in-place last pass x non-temporal first-pass loads, twice" | tee -a $OUT
   for rep in 1 2; do
     for v in "" _ntload2; do
       pp "$v" BDSP_FFT_NO_LAST_INPLACE=1 "lab$v out-of-place" --prec f64 --bits 22 --flags 2 --window 4 0.5
       pp "$v" BDSP_FFT_LAST_INPLACE=1 "lab$v in-place" --prec f64 --bits 22 --flags 2 --window 4 0.5
     done
   done
   echo "## I'. the same per pass (rocprofv3 --kernel-trace, the COLD calls only: tools/trace_tail.py)" | tee -a $OUT
   for v in "" _ntload2; do for ip in BDSP_FFT_NO_LAST_INPLACE BDSP_FFT_LAST_INPLACE; do
     rm -rf gpurun_out/prof_c4a; export BDSP_HIP_LIBRARY=$L$v.so; export $ip=1
     rocprofv3 --kernel-trace -d gpurun_out/prof_c4a -o t --output-format csv -- python3 tools/plan_probe.py --only cold --iters 20 --prec f64 --bits 22 --flags 2 --window 4 0.5 > /dev/null 2>&1
     unset $ip BDSP_HIP_LIBRARY
     echo "lab$v $ip" | tee -a $OUT
     python3 tools/trace_tail.py gpurun_out/prof_c4a 20 k_fft_pass | tee -a $OUT
   done; done ;;
J) echo "## J. cold = input AND scratch cold (one scratch per call): in-place last pass, f32 2^21 / 2^22 plans, C4a; twice" | tee -a $OUT
   for rep in 1 2; do
     for p in f32 f64; do
       pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec $p --bits 17,19,20,21,22
       pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec $p --bits 17,19,20,21,22
     done
     pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec f32 --bits 22 --plans default,256x16:128x32:128x32,4096x4:1024x8,1024x8:4096x4
     pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec f32 --bits 22 --plans default,4096x4:1024x8,1024x8:4096x4
     pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec f32 --bits 21 --plans default,2048x4:1024x8,128x32:128x32:128x32
     pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec f32 --bits 21 --plans default,2048x4:1024x8
     pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec f64 --bits 22 --plans default,256x16:128x32:128x32,2048x2:2048x4
     pp "" BDSP_FFT_NO_LAST_INPLACE=1 out-of-place --prec f64 --bits 20 --batch 16
     pp "" BDSP_FFT_LAST_INPLACE=1 in-place --prec f64 --bits 20 --batch 16
     for v in "" _ntload2; do
       pp "$v" BDSP_FFT_NO_LAST_INPLACE=1 "lab$v out-of-place" --prec f64 --bits 22 --flags 2 --window 4 0.5 --plans default,256x16:128x32:128x32
       pp "$v" BDSP_FFT_LAST_INPLACE=1 "lab$v in-place" --prec f64 --bits 22 --flags 2 --window 4 0.5
     done
   done ;;
K) echo "## K. non-temporal loads in the FIRST pass only, across sizes (the library's own plans, last pass in place from 2^19 on)" | tee -a $OUT
   for rep in 1 2; do for v in "" _ntload2; do for p in f32 f64; do
     pp "$v" - "lab$v" --prec $p --bits 18,19,20,21,22,23
   done; done; done
   for v in "" _ntload2; do
     pp "$v" - "lab$v" --prec f64 --bits 20 --batch 16
     pp "$v" - "lab$v" --prec f32 --bits 20 --flags 8
     pp "$v" - "lab$v" --prec f64 --bits 21,22 --flags 2 --window 4 0.5
   done ;;
L) echo "## L. non-temporal first-pass loads where the first pass reads whole 128-byte lines (3-pass sizes, batches), twice" | tee -a $OUT
   for rep in 1 2; do for v in "" _ntload2; do
     pp "$v" - "lab$v" --prec f32 --bits 23,24,25
     pp "$v" - "lab$v" --prec f64 --bits 23,24
     pp "$v" - "lab$v" --prec f32 --bits 20 --batch 16
     pp "$v" - "lab$v" --prec f32 --bits 20 --batch 64
     pp "$v" - "lab$v" --prec f32 --bits 16 --batch 256
     pp "$v" - "lab$v" --prec f32 --bits 14 --batch 4096
     pp "$v" - "lab$v" --prec f64 --bits 20 --batch 4
     pp "$v" - "lab$v" --prec f64 --bits 20 --batch 16
     pp "$v" - "lab$v" --prec f64 --bits 20 --batch 32
     pp "$v" - "lab$v" --prec f64 --bits 16 --batch 256
     pp "$v" - "lab$v" --prec f64 --bits 22 --batch 4
   done; done
   echo "## L'. is an uploaded vector cold?  (tools/upload_probe.py through the facade)" | tee -a $OUT
   for v in "" _ntload2; do BDSP_HIP_LIBRARY=$L$v.so timeout 300 python3 tools/upload_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT; done ;;
esac; done
date >> $OUT
