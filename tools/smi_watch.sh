#!/bin/bash
# Diagnostic: sample power / clocks while the benchmark runs:  bash tools/smi_watch.sh
(python bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/smi_bench.json 2> /dev/null) &
BP=$!
sleep 25
for i in $(seq 1 12); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use" | tr '\n' ' '
  echo
  sleep 1
done
wait $BP
cat gpurun_out/smi_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
