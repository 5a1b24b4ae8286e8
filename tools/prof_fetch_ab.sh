#!/bin/bash
# FETCH_SIZE of the bench kernels with a set of debug keys: bash tools/prof_fetch_ab.sh "5=0" tagname
set -e
ROOT=$(pwd); P=$ROOT/gpurun_out/prof; mkdir -p $P
cd /tmp && export TMPDIR=/tmp
export INR_DEBUG_KEYS="$1"
rm -rf $P/$2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/$2 -o pmc -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $P/$2.log 2>&1
python3 $ROOT/tools/pmc_summary.py $P/$2 gemm_h3
