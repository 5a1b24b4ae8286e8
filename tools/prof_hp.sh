#!/bin/bash
# GPU box, repo root: in-kernel stamps of the hp GEMMs, then SQ counters of a short bench (gpurun_out/prof/...)
ROOT=$(pwd)
INR_LIB=$ROOT/mri-super-resolution_amd/libinrhip_diag.so bash -c 'for w in fwd dx dw; do python tools/hp_stamps.py $w 1; done' > gpurun_out/hp_stamps.txt 2>&1
bash tools/prof_pmc.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" pmc_sq > gpurun_out/prof_sq.txt 2>&1
true
