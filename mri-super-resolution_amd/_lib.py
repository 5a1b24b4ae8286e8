"""ctypes binding of libinrhip.so (include/inrhip.h).  No compute happens here.

The library is REQUIRED: there is no CPU or PyTorch fallback for the kernels.  ``lib()`` raises
``InrHipUnavailable`` if the shared object has not been built.
"""
from __future__ import annotations

import ctypes as C
import os

from ._build import LIB_PATH

c_f32p = C.c_void_p      # raw device pointers travel as integers (tensor.data_ptr())
c_i64p = C.POINTER(C.c_int64)
c_stream = C.c_void_p


INR_E_INVALID, INR_E_WORKSPACE, INR_E_ALIGN, INR_E_TIMEOUT = -1, -2, -3, -4     # include/inrhip.h
INR_LF_COUNT = 13


class InrHipError(RuntimeError):
    """A libinrhip.so entry point returned a non-zero status."""


class InrHipUnavailable(RuntimeError):
    """libinrhip.so is missing (not built) -- the HIP path cannot run and nothing replaces it."""


class SirenDesc(C.Structure):
    _fields_ = [("in_features", C.c_int), ("hidden_features", C.c_int), ("hidden_layers", C.c_int),
                ("out_features", C.c_int), ("first_omega", C.c_float), ("hidden_omega", C.c_float)]


class RamsDesc(C.Structure):
    _fields_ = [("scale", C.c_int), ("filters", C.c_int), ("kernel_size", C.c_int), ("channels", C.c_int),
                ("r", C.c_int), ("n_rfab", C.c_int), ("mean", C.c_float), ("std", C.c_float)]


class DeviceCaps(C.Structure):
    _fields_ = [("abi_version", C.c_int), ("device", C.c_int), ("compute_units", C.c_int),
                ("wavefront_size", C.c_int), ("lds_bytes_per_cu", C.c_int), ("clock_khz", C.c_int),
                ("hbm_bytes", C.c_int64), ("arch", C.c_char * 32)]


# name -> (restype, argtypes); must list EVERY symbol declared in include/inrhip.h
SIGNATURES = {
    "inr_version": (C.c_int, []),
    "inr_build_flags": (C.c_int, []),
    "inr_last_error": (C.c_char_p, []),
    "inr_device_caps": (C.c_int, [C.c_int, C.POINTER(DeviceCaps)]),
    "inr_mgrid": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int64, C.c_int64, c_stream]),
    "inr_fourier_map": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int, c_stream]),
    "inr_grid_fourier_map": (C.c_int, [c_f32p, c_i64p, C.c_int, C.c_int64, C.c_int64, c_f32p, C.c_int, c_stream]),
    "inr_sine_layer_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int,
                                         C.c_float, c_stream]),
    "inr_tanh_layer_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int,
                                         C.c_float, c_stream]),
    "inr_linear_tanh_head_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int,
                                               C.c_float, c_stream]),
    "inr_mul": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, c_stream]),
    "inr_linear_head_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                          C.c_float, c_stream]),
    "inr_mse_workspace_bytes": (C.c_size_t, [C.c_int64]),
    "inr_mse_loss_grad": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_size_t,
                                    c_stream]),
    "inr_head_backward_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "inr_linear_head_backward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64,
                                           C.c_int, C.c_int, C.c_void_p, C.c_size_t, c_stream]),
    "inr_sine_layer_backward_input_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "inr_sine_layer_backward_input": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int,
                                                C.c_void_p, C.c_size_t, c_stream]),
    "inr_linear_param_grad_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "inr_linear_param_grad": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                        C.c_size_t, c_stream]),
    "inr_adam_step": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int64, C.c_double, C.c_double,
                                C.c_double, C.c_double, c_stream]),
    "inr_siren_param_count": (C.c_int64, [C.POINTER(SirenDesc)]),
    "inr_siren_param_offsets": (C.c_int, [C.POINTER(SirenDesc), c_i64p]),
    "inr_siren_forward_workspace_bytes": (C.c_size_t, [C.POINTER(SirenDesc), C.c_int64]),
    "inr_siren_forward": (C.c_int, [C.POINTER(SirenDesc), c_f32p, c_f32p, C.c_int64, c_f32p, C.c_int, C.c_float,
                                    C.c_void_p, C.c_size_t, c_stream]),
    "inr_siren_reconstruct_workspace_bytes": (C.c_size_t, [C.POINTER(SirenDesc), C.c_int64]),
    "inr_siren_reconstruct": (C.c_int, [C.POINTER(SirenDesc), c_f32p, c_i64p, C.c_int, c_f32p, C.c_int, c_f32p,
                                        C.c_int, C.c_float, C.c_int64, C.c_void_p, C.c_size_t, c_stream]),
    "inr_siren_fit_workspace_bytes": (C.c_size_t, [C.POINTER(SirenDesc), C.c_int64]),
    "inr_siren_fit": (C.c_int, [C.POINTER(SirenDesc), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "inr_siren_fit_cycle": (C.c_int, [C.POINTER(SirenDesc), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int,
                                      C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double,
                                      C.c_double, c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "inr_metric_workspace_bytes": (C.c_size_t, [C.c_int]),
    "inr_psnr": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_int64, C.c_double, C.c_void_p, C.c_size_t, c_stream]),
    "inr_ssim2d": (C.c_int, [C.c_void_p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                             C.c_float, C.c_void_p, C.c_size_t, c_stream]),
    "inr_rams_train_param_count": (C.c_int64, [C.POINTER(RamsDesc)]),
    "inr_rams_train_param_offsets": (C.c_int, [C.POINTER(RamsDesc), c_i64p, C.c_int]),
    "inr_rams_train_workspace_bytes": (C.c_size_t, [C.POINTER(RamsDesc), C.c_int, C.c_int, C.c_int]),
    "inr_rams_train_grads": (C.c_int, [C.POINTER(RamsDesc), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p, c_f32p, C.c_int,
                                       C.c_int, C.c_int, C.c_void_p, C.c_size_t, c_stream]),
    "inr_rams_train_step": (C.c_int, [C.POINTER(RamsDesc), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double,
                                      C.c_void_p, C.c_size_t, c_stream]),
    "inr_acquisition_products": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, C.c_int, C.c_int,
                                           c_stream]),
    "inr_rescale2d_linear": (C.c_int, [c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_stream]),
    "inr_adc_map": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int, c_stream]),
    "inr_rams_param_count": (C.c_int64, [C.POINTER(RamsDesc)]),
    "inr_rams_workspace_bytes": (C.c_size_t, [C.POINTER(RamsDesc), C.c_int, C.c_int, C.c_int]),
    "inr_rams_forward": (C.c_int, [C.POINTER(RamsDesc), c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p, C.c_size_t, c_stream]),
    "inr_siren_loss_grad": (C.c_int, [C.POINTER(SirenDesc), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int64,
                                      c_f32p, C.c_void_p, C.c_size_t, c_stream]),
    "inr_siren_loss_grad_ex": (C.c_int, [C.POINTER(SirenDesc), c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int64, C.c_int64,
                                         c_f32p, C.c_void_p, C.c_size_t, C.c_int, c_stream]),
    "inr_siren_hp_eligible": (C.c_int, [C.POINTER(SirenDesc)]),
    "inr_siren_forward_train": (C.c_int, [C.POINTER(SirenDesc), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_size_t, C.c_int,
                                          c_stream]),
    "inr_siren_backward_train": (C.c_int, [C.POINTER(SirenDesc), c_f32p, c_f32p, c_f32p, C.c_int64, C.c_void_p, C.c_size_t,
                                           c_stream]),
    "inr_rams_shift_loss_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "inr_rams_shift_loss": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_size_t, c_stream]),
    "inr_rams_conv3d_forward": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          c_stream]),
    "inr_rams_conv3d_dgrad_workspace_bytes": (C.c_size_t, []),
    "inr_rams_conv3d_dgrad": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t,
                                        c_stream]),
    "inr_rams_conv3d_wgrad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "inr_rams_conv3d_wgrad": (C.c_int, [c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_size_t, c_stream]),
    "inr_rams_shift_loss_grad_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "inr_rams_shift_loss_grad": (C.c_int, [C.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_int, C.c_int, C.c_int,
                                           C.c_void_p, C.c_size_t, c_stream]),
    "inr_hybrid_fit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, c_stream]),
    "inr_auto_erd": (C.c_int, [c_f32p, C.c_void_p, c_f32p, C.c_int64, C.c_int, C.c_int, c_stream]),
    "inr_prof_enable": (C.c_int, [C.c_int]),
    "inr_prof_reset": (C.c_int, []),
    "inr_prof_read": (C.c_int, [C.c_int, c_i64p, C.POINTER(C.c_double)]),
    "inr_debug_set": (C.c_int, [C.c_int, C.c_int]),
    "inr_debug_get": (C.c_int, [C.c_int, C.POINTER(C.c_int)]),
    "inr_debug_reset": (C.c_int, []),
    "inr_launch_count": (C.c_int, [C.c_int, c_i64p]),
    "inr_launch_counts_reset": (C.c_int, []),
    "inr_debug_set_ptr": (C.c_int, [C.c_int, C.c_void_p]),
    "inr_sincos_probe": (C.c_int, [c_f32p, c_f32p, c_f32p, C.c_int64, c_stream]),
}

_LIB = None


def lib():
    """The loaded library with argtypes set.  Raises InrHipUnavailable when it is not built."""
    global _LIB
    if _LIB is None:
        path = os.environ.get("INR_LIB") or LIB_PATH       # INR_LIB: an explicitly selected (diagnostic) build
        if not os.path.exists(path):
            raise InrHipUnavailable(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no fallback path.")
        handle = C.CDLL(path)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        if handle.inr_version() != 1:
            raise InrHipUnavailable(f"libinrhip.so ABI version {handle.inr_version()} != 1: rebuild")
        if handle.inr_build_flags() != 0 and not os.environ.get("INR_LIB"):
            raise InrHipUnavailable(
                f"{path} is a DIAGNOSTIC build (inr_build_flags() = {handle.inr_build_flags()}: time stamps / ablated "
                "kernels); rebuild the product library (`python mri-super-resolution_amd/_build.py --force`) or select the "
                "diagnostic one explicitly with INR_LIB=<path>")
        _LIB = handle
    return _LIB


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().inr_last_error().decode("utf-8", "replace")
        raise InrHipError(f"{what or 'libinrhip'} failed with status {rc}: {msg}")


def shape_array(shape):
    arr = (C.c_int64 * len(shape))(*[int(s) for s in shape])
    return arr
