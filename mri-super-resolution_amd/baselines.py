"""The interpolation baselines the reference reports its results against, on the device.

``rescale`` is ``skimage.transform.rescale(image, scale, anti_aliasing=True)`` as called at superresDWI.py:172-191 and
master.py:175 for up-scaling 2-D images: default order 1, mode 'reflect' -- which skimage 0.20 evaluates as
``scipy.ndimage.zoom(image, scale, order=1, mode='mirror', grid_mode=True)``; the anti-aliasing Gaussian has sigma
``max(0, (1/scale - 1)/2) = 0`` when ``scale >= 1``.  scikit-image itself is not installed in the build image: parity with
skimage is therefore UNPINNED; the kernel is pinned against the scipy call skimage makes (tests).  Down-scaling (a real
anti-aliasing filter) is not needed by the drivers and is rejected.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import check, lib


def rescale(image, scale, anti_aliasing: bool = True):
    """Up-scales the trailing two axes of ``image`` (ndarray or device tensor; leading axes are a batch) by ``scale``.
    Returns the type it was given (ndarray in -> float64 ndarray out, like skimage; tensor in -> fp32 device tensor)."""
    if scale < 1:
        raise ValueError("rescale: only up-scaling (scale >= 1) is implemented -- the drivers never down-scale")
    as_numpy = isinstance(image, np.ndarray)
    x = torch.from_numpy(np.ascontiguousarray(image, dtype=np.float32)).to(ops.require_gpu()) if as_numpy else image
    ops._chk(x, "image")
    if x.dim() < 2:
        raise ValueError("rescale needs at least 2-D input")
    h, w = x.shape[-2], x.shape[-1]
    oh, ow = int(round(h * scale)), int(round(w * scale))        # skimage: np.round(scale * input_shape)
    flat = x.reshape(-1, h, w).contiguous()
    out = torch.empty((flat.shape[0], oh, ow), dtype=torch.float32, device=x.device)
    check(lib().inr_rescale2d_linear(out.data_ptr(), flat.data_ptr(), flat.shape[0], h, w, oh, ow, ops._stream()),
          "inr_rescale2d_linear")
    out = out.reshape(*x.shape[:-2], oh, ow)
    return out.cpu().numpy().astype(np.float64) if as_numpy else out
