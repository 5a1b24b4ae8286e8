#!/bin/bash
# GPU box: RAMS training step (batch 32 of 32x32x9 patches) under rocprofv3 --kernel-trace --stats -> gpurun_out/prof/rams_train_kernel_stats.csv
set -e
ROOT=$(pwd)
P=$ROOT/gpurun_out/prof
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rm -rf $P/rams_train_kt
rocprofv3 --kernel-trace --stats --output-format csv -d $P/rams_train_kt -o kt -- python3 $ROOT/tools/rams_train_time.py > $P/rams_train_kt.log 2>&1
cp $(find $P/rams_train_kt -name "*kernel_stats.csv" | head -1) $P/rams_train_kernel_stats.csv
tail -3 $P/rams_train_kt.log
python3 $ROOT/tools/kt_summary.py $P/rams_train_kt 6 16
