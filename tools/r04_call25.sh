#!/bin/bash
# round 4, after the code-generation changes: smoke, the GPU suite, then the profile passes and the step-time table
set -o pipefail
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/r4_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4_smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t25.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r4_t25.log
[ $rc -eq 0 ] || exit $rc
bash tools/prof_all.sh > gpurun_out/r4_prof_all.log 2>&1; echo "prof_all rc=$?"; tail -4 gpurun_out/r4_prof_all.log
timeout -k 10 400 python tools/step_time_table.py gpurun_out/r04_step_time_table.json > gpurun_out/r4_step_table.log 2>&1; tail -3 gpurun_out/r4_step_table.log
