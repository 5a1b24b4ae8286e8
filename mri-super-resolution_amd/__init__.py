"""mri-super-resolution_amd -- MI355X (gfx950) implementation of the INR / SIREN super-resolution
fit path of MRIRC/MRI-super-resolution.

The directory name carries a hyphen (it is the project name), so import it through the alias package
``mri_super_resolution_amd`` at the repo root, or put ``mri-super-resolution_amd/compat`` on
``sys.path`` to get drop-in ``SRDWI`` / ``INRmodel`` / ``nn_mri`` modules for the reference's drivers.

Layout:  csrc/ (HIP kernels + C ABI, built into libinrhip.so) . _lib.py (ctypes binding) . ops.py
(tensor wrappers) . inr.py (reference module surface + fused fit / reconstruct) . metrics.py . baselines.py (spline
rescale) . drivers.py (the reference's driver loops) . dist.py (fit partitioning over GPUs, metric gather) . matio.py
(.mat level 5) . reports.py (CSV schemas) . contrast.py (case / calculate_contrast) . scripts/ (superresDWI, master).
"""
from ._lib import InrHipError, InrHipUnavailable  # noqa: F401
from .ops import InrDeviceError  # noqa: F401
from .inr import (ImageFitting_set, PN, ShardedSirenFitter, SineLayer, Siren, SirenFitter, calculate_ADC,  # noqa: F401
                  calculate_combinations, fit_siren, flat_parameters, get_mgrid, input_mapping, reconstruct,
                  resize_array)

__all__ = ["ImageFitting_set", "PN", "ShardedSirenFitter", "SineLayer", "Siren", "SirenFitter", "calculate_ADC",
           "calculate_combinations", "fit_siren", "flat_parameters", "get_mgrid", "input_mapping",
           "reconstruct", "resize_array", "InrHipError", "InrHipUnavailable", "InrDeviceError"]
