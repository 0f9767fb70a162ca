#!/bin/bash
# run the edge block in a loop for ~15 s and sample clocks / power meanwhile
python tools/fusedbench.py --iters 3000 > gpurun_out/r3b/loop.log 2>&1 &
PID=$!
sleep 6
for i in 1 2 3 4 5; do
  rocm-smi --showclocks --showpower -d 0 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" >> gpurun_out/r3b/smi.log
  echo "--" >> gpurun_out/r3b/smi.log
  sleep 1.5
done
wait $PID
echo "idle:" >> gpurun_out/r3b/smi.log
sleep 3
rocm-smi --showclocks --showpower -d 0 2>/dev/null | grep -E "sclk|mclk|fclk|Power|power" >> gpurun_out/r3b/smi.log
