#!/bin/bash
# GPU box: bash tools/prof_rams_pmc.sh [key15, default 42 = the product default]: SQ counters of the RAMS forward at batch 25 (one pass) -> gpurun_out/prof/rams_pmc
set -e
ROOT=$(pwd)
P=$ROOT/gpurun_out/prof
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rm -rf $P/rams_pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $P/rams_pmc -o pmc -- python3 $ROOT/tools/rams_prof.py 25 ${1:-42} > $P/rams_pmc.log 2>&1
python3 $ROOT/tools/pmc_summary.py $P/rams_pmc conv3d_c32_lds
