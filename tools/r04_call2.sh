#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_autograd_hp.py tests/test_gpu_drivers.py tests/test_gpu_parity.py -x -q > gpurun_out/r4_t2.log 2>&1; echo "tests rc=$?"
tail -15 gpurun_out/r4_t2.log
timeout -k 10 300 python tools/compat_time.py > gpurun_out/r4_compat_time.txt 2>&1; cat gpurun_out/r4_compat_time.txt | tail -5
