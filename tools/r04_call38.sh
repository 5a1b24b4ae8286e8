#!/bin/bash
# round 4, end: smoke, the GPU suite, the profile passes, RAMS traces, the full bench line
set -o pipefail
mkdir -p gpurun_out
python __graft_entry__.py smoke > gpurun_out/r4_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r4_smoke.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r4_t38.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r4_t38.log
[ $rc -eq 0 ] || exit $rc
bash tools/prof_all.sh > gpurun_out/r4_prof_all.log 2>&1; echo "prof_all rc=$?"; tail -4 gpurun_out/r4_prof_all.log
bash tools/prof_rams.sh > gpurun_out/r4_prof_rams.log 2>&1; tail -2 gpurun_out/r4_prof_rams.log
bash tools/prof_rams_train.sh > gpurun_out/r4_prof_rams_train.log 2>&1; tail -3 gpurun_out/r4_prof_rams_train.log
