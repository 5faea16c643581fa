#!/bin/bash
# Samples the GPU clock and power while the convolution kernel runs in a loop (run through gpurun).
python3 tools/kbench.py --what conv --iters 300000 > /tmp/kb.log 2>&1 &
PID=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|power" | head -6
  sleep 0.5
done
wait $PID
tail -1 /tmp/kb.log
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | head -4
