"""AutoERD acceptance weights of the 2-D slice driver (master.py:77-93) on the device.

The reference loops over the ROI in Python and fits one ``sklearn.cluster.AgglomerativeClustering(n_clusters=2,
linkage='complete')`` per pixel (3,600 fits per case) to mark outlying acquisitions in ``case.accept``; ``inr_auto_erd`` does
all pixels in one launch with the same partition, ties included (csrc/metrics.hip).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import check, lib


def auto_erd(img, rule: int, erd_map=None) -> np.ndarray:
    """``img`` [H, W, n_acquisitions] (all acquisitions of one slice, any real dtype) -> accept [H, W, n] of 0 / 1 (int64, the
    dtype of ``case.accept``).  ``rule`` 1 = majority voting (--erd 1), 2 = intensity-cognisant (--erd 2; ``erd_map`` [H, W]:
    pixels where it is not positive keep everything, master.py:89)."""
    dev = ops.require_gpu()
    a = np.ascontiguousarray(np.asarray(img), dtype=np.float64)
    if a.ndim != 3:
        raise ValueError("img must be [H, W, n_acquisitions]")
    H, W, n = a.shape
    vals = torch.from_numpy(a.reshape(H * W, n)).to(dev)
    emap = None
    if erd_map is not None:
        e = np.ascontiguousarray(np.asarray(erd_map), dtype=np.float32)
        if e.shape != (H, W):
            raise ValueError(f"erd_map must be [{H}, {W}]")
        emap = torch.from_numpy(e.reshape(-1)).to(dev)
    out = torch.empty((H * W, n), dtype=torch.float32, device=dev)
    check(lib().inr_auto_erd(out.data_ptr(), vals.data_ptr(), 0 if emap is None else emap.data_ptr(), H * W, n, int(rule),
                             torch.cuda.current_stream().cuda_stream), "inr_auto_erd")
    return out.cpu().numpy().reshape(H, W, n).astype(np.int64)


def apply_auto_erd(case, rule: int, roi_begin: int, roi_end: int) -> None:
    """master.py:77-93 on a ``contrast.case``: clears ``case.accept`` inside the ROI of the cancer slice for the rejected
    acquisitions (rule 2 needs ``case.erd``)."""
    s = case.cancer_slice
    img = case.dwi[roi_begin:roi_end, roi_begin:roi_end, s, :]
    emap = None
    if rule == 2:
        if case.erd is None:
            raise ValueError("--erd 2 needs the patient's ERD map (pat<NN>_ERD.mat: ADC_alldata_mm_ERD)")
        emap = case.erd[roi_begin:roi_end, roi_begin:roi_end, s]
    keep = auto_erd(img, rule, emap)
    block = case.accept[roi_begin:roi_end, roi_begin:roi_end, s, :]
    block[keep == 0] = 0
