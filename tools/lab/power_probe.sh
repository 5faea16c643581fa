#!/bin/bash
# Power and clock of the chip while ONE lab variant runs for several seconds (run through gpurun).
# usage: tools/lab/power_probe.sh "<variant name>" [binary]
BIN=${2:-tools/lab/conv_lab}
LAB_SECONDS=5 $BIN 16777216 1024 "$1" > /tmp/lab_$$.log 2>&1 &
PID=$!
sleep 5.5
for i in 1 2 3; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|fclk|mclk|Power|power" | tr '\n' ' '; echo
  sleep 0.6
done
wait $PID
cut -c1-140 /tmp/lab_$$.log
