#!/bin/bash
# GPU box: RAMS forward (batch 1 / 5 / 25) on the product library and on diagnostic builds of the LDS-staged convolution:
#   libinrhip_r3abl<N>.so  -DR3_ABLATE=<N>: 2 = no MFMAs, 8 = no weight loads, 16 = no LDS operand reads, 24 = neither,
#                          32 = no output stores, 64 = no staging, 96 = neither stores nor staging
#   libinrhip_skew<N>.so   -DR3_SKEW=<N>: head start (x 64 cycles) of the first wave of each SIMD
# Build them first (no GPU needed):  python tools/rams_ablate_build.py abl 8 16 ... / skew 0 28 100
ROOT=$(pwd)
unset INR_LIB
echo "== product"; python tools/rams_timing.py 2>&1 | grep "RAMS forward"
for lib in $ROOT/mri-super-resolution_amd/libinrhip_r3abl*.so $ROOT/mri-super-resolution_amd/libinrhip_skew*.so; do
  [ -f "$lib" ] || continue
  export INR_LIB=$lib
  echo "== $(basename $lib)"; python tools/rams_timing.py 2>&1 | grep "RAMS forward"
done
