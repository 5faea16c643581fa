#!/bin/bash
# Power and clock of the chip while ONE lab variant runs for several seconds (run through gpurun).
# usage: tools/lab/power_probe.sh "<variant name>" [binary]
BIN=${2:-tools/lab/conv_lab}
LAB_SECONDS=6 $BIN 16777216 1024 "$1" > /tmp/lab_$$.log 2>&1 &
PID=$!
sleep 4.0   # host set-up (reference outputs, uploads) takes ~2 s, then the timed loop, then the long run
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|power \(W\)|Package Power" | sed 's/^GPU\[0\][ \t]*: //' | tr '\n' ';'; echo
  sleep 0.8
done
wait $PID
grep -E "sustained|wg/CU" /tmp/lab_$$.log | cut -c1-110
