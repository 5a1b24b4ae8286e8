"""End-of-fit metrics on the device, through libinrhip.so.

``psnr`` = ``10*log10(data_range**2 / MSE)`` (skimage's ``peak_signal_noise_ratio``, imported but unused at
master.py:14 -- the build adds it because north_star asks for PSNR); ``ssim`` follows skimage 0.20
``structural_similarity`` defaults as called at superresDWI.py:186; ``ssim_reference_protocol`` is the whole
per-slice recipe of superresDWI.py:179-186 (max-normalise both, mask by HR > 0.05, data_range = 1);
``calculate_ADC_device`` is ``calculate_ADC`` (SRDWI.py:118-130) for a device-resident stack.
Inputs are device fp32 tensors; results are float64 device tensors (no host sync inside).
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import check, lib


def _as_images(x: torch.Tensor, name: str):
    ops._chk(x, name)
    if x.dim() < 2:
        raise ValueError(f"{name} must be at least 2-D")
    h, w = x.shape[-2], x.shape[-1]
    return x.reshape(-1, h, w), h, w


def psnr(ref: torch.Tensor, test: torch.Tensor, data_range: float = 1.0, per_image: bool = False) -> torch.Tensor:
    """PSNR of ``test`` against ``ref``; one value over everything, or one per leading-index image."""
    ops._chk(ref, "ref")
    ops._chk(test, "test", ref.shape)
    nimg = ref.reshape(-1, ref.shape[-2] * ref.shape[-1]).shape[0] if per_image and ref.dim() >= 2 else 1
    n_per = ref.numel() // nimg
    out = torch.empty(nimg, dtype=torch.float64, device=ref.device)
    ws = ops._ws(lib().inr_metric_workspace_bytes(nimg), ref.device)
    check(lib().inr_psnr(out.data_ptr(), ref.data_ptr(), test.data_ptr(), nimg, n_per, float(data_range),
                         ws.data_ptr(), ws.numel(), ops._stream()), "inr_psnr")
    return out if per_image else out[0]


def ssim(im1: torch.Tensor, im2: torch.Tensor, data_range: float = 1.0, win_size: int = 7, mask_threshold=None):
    """SSIM per 2-D image (trailing two dims); leading dims are a batch.  ``mask_threshold`` multiplies both
    images by ``im1 > threshold`` first."""
    x, h, w = _as_images(im1, "im1")
    y, _, _ = _as_images(im2, "im2")
    if x.shape != y.shape:
        raise ValueError("im1 / im2 shape mismatch")
    if win_size % 2 == 0 or win_size < 3 or h < win_size or w < win_size:
        raise ValueError("win_size must be odd, >= 3 and not larger than the image")
    nimg = x.shape[0]
    out = torch.empty(nimg, dtype=torch.float64, device=x.device)
    ws = ops._ws(lib().inr_metric_workspace_bytes(nimg), x.device)
    check(lib().inr_ssim2d(out.data_ptr(), x.data_ptr(), y.data_ptr(), nimg, h, w, int(win_size), float(data_range),
                           0 if mask_threshold is None else 1, float(mask_threshold or 0.0), ws.data_ptr(), ws.numel(),
                           ops._stream()), "inr_ssim2d")
    return out.reshape(im1.shape[:-2]) if im1.dim() > 2 else out[0]


def ssim_reference_protocol(hr: torch.Tensor, sr: torch.Tensor, mask_threshold: float = 0.05) -> torch.Tensor:
    """superresDWI.py:179-186 per 2-D slice: HR/HR.max(), SR/SR.max(), mask = HR > 0.05,
    ``ssim(HR*mask, SR*mask, data_range=1)``."""
    hr_n = (hr / hr.amax(dim=(-2, -1), keepdim=True)).contiguous()
    sr_n = (sr / sr.amax(dim=(-2, -1), keepdim=True)).contiguous()
    return ssim(hr_n, sr_n, data_range=1.0, win_size=7, mask_threshold=mask_threshold)


def calculate_ADC_device(bvalues, slicedata: torch.Tensor) -> torch.Tensor:
    """``calculate_ADC(bvalues, slicedata)`` for a device tensor ``[..., n_b]``; returns fp32 ``[...]``."""
    ops._chk(slicedata, "slicedata")
    nb = slicedata.shape[-1]
    b = torch.as_tensor(bvalues, dtype=torch.float32).reshape(-1).to(slicedata.device)
    if b.numel() != nb:
        raise ValueError("bvalues length must match the last axis of slicedata")
    out = torch.empty(slicedata.shape[:-1], dtype=torch.float32, device=slicedata.device)
    check(lib().inr_adc_map(out.data_ptr(), slicedata.data_ptr(), b.data_ptr(), out.numel(), nb, ops._stream()),
          "inr_adc_map")
    return out
