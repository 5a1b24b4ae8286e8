#!/bin/bash
# diagnostic copy of the library with in-kernel time stamps, in its OWN file (mri-super-resolution_amd/libinrhip_diag.so);
# use it with INR_LIB=$PWD/mri-super-resolution_amd/libinrhip_diag.so (the binding refuses a diagnostic build otherwise)
set -e
cd "$(dirname "$0")/.."
python mri-super-resolution_amd/_build.py --diag -DINR_STAMPS
