#!/bin/bash
# usage (GPU box, repo root): bash tools/prof_pmc.sh "<counters>" <tag>  -> gpurun_out/prof/<tag>/ (counter collection of a short bench)
set -e
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/prof/$2
rocprofv3 --pmc $1 --output-format csv -d $ROOT/gpurun_out/prof/$2 -o pmc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $ROOT/gpurun_out/prof/$2.log 2>&1
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/prof/$2
