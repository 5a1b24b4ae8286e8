#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t16.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4_t16.log
bash tools/prof_all.sh > gpurun_out/r4_prof_all.log 2>&1; echo "prof_all rc=$?"
