#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 200 python tools/recon_chunks.py > gpurun_out/r4_recon_chunks.txt 2>&1; cat gpurun_out/r4_recon_chunks.txt
timeout -k 10 600 python -m pytest tests/test_gpu_cfg4.py tests/test_gpu_drivers.py tests/test_gpu_rams.py -x -q -k "not t4" > gpurun_out/r4_t8.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r4_t8.log
