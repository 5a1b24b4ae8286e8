#!/bin/bash
# GPU box: shader clock and package power while the fused fit step runs (is the step power-limited?)
python bench.py --steps 2500 --warmup 3 --no-cpu-baseline --no-extras > /tmp/cw_bench.log 2>&1 &
BP=$!
sleep 9
for i in $(seq 1 14); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 1
done
wait $BP
grep -o '"ms_per_step": [0-9.]*' /tmp/cw_bench.log | head -1
