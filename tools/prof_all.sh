#!/bin/bash
# On the GPU box, from the repo root: bash tools/prof_all.sh
# Kernel trace + the three PMC passes (separate runs, as gpurun requires) of the short bench; outputs under gpurun_out/prof/.
# Afterwards, in the build container: python tools/save_profiles.py rNN
set -e
ROOT=$(pwd)
P=$ROOT/gpurun_out/prof
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rm -rf $P/kt $P/pmc_fetch $P/pmc_write $P/pmc_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -o kt -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $P/kt.log 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -o pmc -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $P/f.log 2>&1
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -o pmc -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $P/w.log 2>&1
echo "write pass done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $P/pmc_sq -o pmc -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $P/s.log 2>&1
echo "sq pass done"
