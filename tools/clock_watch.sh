#!/bin/bash
# GPU box: shader clock and package power (rocm-smi, once a second) while a workload runs -- is it power-limited?
#   bash tools/clock_watch.sh                 the fused fit step (2,500 steps)
#   bash tools/clock_watch.sh fp32            the same on the exact-fp32 MFMA kernels
#   bash tools/clock_watch.sh rams            RAMS forward, batch 25, in a loop
case "$1" in
  fp32) python bench.py --fp32-mfma --steps 1200 --warmup 3 --no-cpu-baseline --no-extras > /tmp/cw.log 2>&1 & ;;
  rams) python -c "
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from mri_super_resolution_amd import rams
m = rams.RAMS(seed=0)
x = torch.from_numpy((np.random.default_rng(0).random((25, 128, 128, 9)) * 60000).astype(np.float32)).cuda()
for _ in range(700): m(x)
torch.cuda.synchronize()
" > /tmp/cw.log 2>&1 & ;;
  *) python bench.py --steps 2500 --warmup 3 --no-cpu-baseline --no-extras > /tmp/cw.log 2>&1 & ;;
esac
BP=$!
sleep 10
for i in $(seq 1 8); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //; s/ =* Power Consumption =* / W: /' | tr '\n' ' '; echo
  sleep 1
done
wait $BP
grep -o '"ms_per_step": [0-9.]*' /tmp/cw.log | head -1
