#!/bin/bash
# On the GPU box, from the repo root: bash tools/prof_rams.sh   -> gpurun_out/prof/rams_b25_kernel_stats.csv, rams_b1_...
set -e
ROOT=$(pwd)
P=$ROOT/gpurun_out/prof
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
for B in 25 1; do
  rm -rf $P/rams_kt$B
  rocprofv3 --kernel-trace --stats --output-format csv -d $P/rams_kt$B -o kt -- python3 $ROOT/tools/rams_prof.py $B > $P/rams_kt$B.log 2>&1
  cp $(find $P/rams_kt$B -name "*kernel_stats.csv" | head -1) $P/rams_b${B}_kernel_stats.csv
  echo "batch $B done"
done
