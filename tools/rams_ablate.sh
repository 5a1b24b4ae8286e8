#!/bin/bash
# GPU box: RAMS forward at batch 25 on the product library and on -DR3_ABLATE builds of the LDS-staged convolution
# (2 = no MFMAs, 8 = no weight loads, 16 = no LDS operand reads, 24 = neither, 32 = no output stores); ms per 25 stacks.
ROOT=$(pwd)
for a in "" 2 8 16 24 32; do
  if [ -z "$a" ]; then unset INR_LIB; tag=product; else export INR_LIB=$ROOT/mri-super-resolution_amd/libinrhip_r3abl$a.so; tag="R3_ABLATE=$a"; fi
  [ -n "$a" ] && [ ! -f "$INR_LIB" ] && { echo "$tag: not built"; continue; }
  echo -n "$tag: "; python tools/rams_b25.py 2>&1 | tail -1
done
