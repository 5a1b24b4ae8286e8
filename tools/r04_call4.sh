#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export INR_LIB=$PWD/mri-super-resolution_amd/libinrhip_diag.so
for w in fwd dx; do for nth in 1 2; do timeout -k 10 120 python tools/nt_stamps.py 4096 $w $nth; done; done > gpurun_out/r4_nt_stamps.txt 2>&1
timeout -k 10 120 python tools/nt_stamps.py 16384 fwd 1 >> gpurun_out/r4_nt_stamps.txt 2>&1
unset INR_LIB
cat gpurun_out/r4_nt_stamps.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_small -o small -- python3 $GRAFT_REPO_ROOT/tools/fit_n.py 4096 200 > $GRAFT_REPO_ROOT/gpurun_out/r4_small_prof.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_small | head
python tools/kt_timeline.py gpurun_out/prof_small 100 2>&1 | head -40
