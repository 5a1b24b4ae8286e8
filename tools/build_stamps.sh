#!/bin/bash
# builds a diagnostic copy of the library with in-kernel time stamps enabled (never shipped / never benchmarked)
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DINR_STAMPS -I include -I mri-super-resolution_amd/csrc \
  -o mri-super-resolution_amd/libinrhip.so mri-super-resolution_amd/csrc/api.hip mri-super-resolution_amd/csrc/gemm_f32.hip mri-super-resolution_amd/csrc/kernels.hip mri-super-resolution_amd/csrc/metrics.hip mri-super-resolution_amd/csrc/rams.hip mri-super-resolution_amd/csrc/siren_small.hip mri-super-resolution_amd/csrc/hybrid_fit.hip
