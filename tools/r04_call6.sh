#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for n in 69632 46421 524288; do
  steps=100; [ $n -gt 200000 ] && steps=30
  bash tools/prof_fit_n.sh $n $steps > gpurun_out/r4_fit_$n.txt 2>&1
  cat gpurun_out/r4_fit_$n.txt | head -16
  python tools/kt_timeline.py gpurun_out/prof/fit_$n 20 2>&1 | head -20
done
