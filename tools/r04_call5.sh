#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_rams.py tests/test_gpu_parity.py tests/test_gpu_hp_variants.py tests/test_gpu_autograd_hp.py -x -q > gpurun_out/r4_t5.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4_t5.log
INR_LIB=$PWD/mri-super-resolution_amd/libinrhip_diag.so timeout -k 10 120 python tools/nt_stamps.py 4096 fwd 1 > gpurun_out/r4_nt_stamps2.txt 2>&1
INR_LIB=$PWD/mri-super-resolution_amd/libinrhip_diag.so timeout -k 10 120 python tools/nt_stamps.py 4096 dx 1 >> gpurun_out/r4_nt_stamps2.txt 2>&1
cat gpurun_out/r4_nt_stamps2.txt
timeout -k 10 500 python tools/ab_small.py 4096,16384,46421,69632,114688 "-" "21=4" "21=8" "22=128" "22=192" "22=384" "21=4,22=192" > gpurun_out/r4_ab_small.txt 2>&1; cat gpurun_out/r4_ab_small.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_small2 -o small -- python3 $GRAFT_REPO_ROOT/tools/fit_n.py 4096 200 > $GRAFT_REPO_ROOT/gpurun_out/r4_small_prof2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kt_timeline.py gpurun_out/prof_small2 100 2>&1 | head -20
