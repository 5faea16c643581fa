#!/bin/bash
# Builds the LAB library and its compile-time A/B variants (round 5, tools/plan_matrix.sh):
#   libbasic_dsp_hip_lab.so          the product's sources + the run-time experiment switches
#   ..._lab_nosplit.so               f64 tiles cross LDS as whole complex values (round 2's exchange; -DBDSP_FFT_NO_SPLIT)
#   ..._lab_ntload2.so               non-temporal loads in the FIRST pass only (-DBDSP_FFT_NTLOAD=2)
#   ..._lab_nt.so                    non-temporal stores in every pass (-DBDSP_FFT_NT)
# Only the FFT translation units differ, so the variants reuse the lab build's other objects.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/basic_dsp_amd/csrc"
J=${J:-8}
make -j$J lab > /tmp/build_lab.log 2>&1 || { tail -30 /tmp/build_lab.log; exit 1; }
for v in "nosplit:-DBDSP_FFT_NO_SPLIT" "ntload2:-DBDSP_FFT_NTLOAD=2" "nt:-DBDSP_FFT_NT"; do
  name=${v%%:*}; def=${v#*:}
  rm -rf build_lab_$name; mkdir -p build_lab_$name
  for o in build_lab/*.o build_lab/exports.map; do
    case $o in */fft_f32.o|*/fft_f64.o) ;; *) cp -p $o build_lab_$name/;; esac
  done
  make -j$J BUILD=build_lab_$name OUT=../lib/libbasic_dsp_hip_lab_$name.so EXTRA="-DBDSP_LAB $def" all > /tmp/build_lab_$name.log 2>&1 || { tail -30 /tmp/build_lab_$name.log; exit 1; }
done
# ... and one whose block kernel reads its input with non-temporal loads (-DBDSP_CONV_NTL): only conv_v2.o differs
rm -rf build_lab_ntl; mkdir -p build_lab_ntl
for o in build_lab/*.o build_lab/exports.map; do case $o in */conv_v2.o) ;; *) cp -p $o build_lab_ntl/;; esac; done
make -j$J BUILD=build_lab_ntl OUT=../lib/libbasic_dsp_hip_lab_ntl.so EXTRA="-DBDSP_LAB -DBDSP_CONV_NTL" all > /tmp/build_lab_ntl.log 2>&1 || { tail -30 /tmp/build_lab_ntl.log; exit 1; }
ls -la ../lib
