"""Three-compartment hybrid fit on the device.

Mirrors the reference's `PIA.three_compartment_fit` (PIA.py:240-251) and `PIA.hybrid_fit` (PIA.py:253-283, called at
superresHybrid.py:140): same argument meaning, same return triple, same failure behaviour (a voxel whose fit exhausts
`maxfev` gets p0; non-finite input raises ValueError as `curve_fit(check_finite=True)` does).  The per-voxel Python
loop around scipy's `curve_fit` becomes one kernel launch (`inr_hybrid_fit`, one voxel per lane, fp64).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import check, lib

P0 = (0.55, 1.3, 2.8, 50.0, 70.0, 750.0, 0.3, 0.4)     # PIA.py:269
LB = (0.3, 0.7, 2.7, 20.0, 40.0, 500.0, 0.0, 0.0)      # PIA.py:271
UB = (0.7, 1.7, 3.0, 70.0, 100.0, 1000.0, 1.0, 1.0)    # PIA.py:272
BVALS = (0.0, 150.0, 1000.0, 1500.0)                   # PIA.py:254
NORM_TE = (0.0, 13.0, 93.0, 143.0)                     # PIA.py:255


def acquisition_table():
    """``xdata`` of PIA.py:263-264 (``meshgrid(normTE, bvals)``: b is the slow index): ``(b [16], TE [16])``."""
    return np.repeat(np.asarray(BVALS), 4), np.tile(np.asarray(NORM_TE), 4)


def phantom_signals(n, noise=0.02, seed=0):
    """Seeded three-compartment signals ``[n, 16]`` in the reference's units, for benchmarks and demos: parameters uniform inside
    the fit bounds, volume fractions normalised, additive Gaussian noise (the recipe of ``PIA.get_batch``, PIA.py:171-213, with
    numpy's own generator)."""
    rng = np.random.default_rng(seed)
    p = np.column_stack([rng.uniform(LB[k], UB[k], n) for k in range(6)])
    vol = rng.uniform(0, 1, (n, 3))
    vol /= vol.sum(axis=1, keepdims=True)
    b, te = acquisition_table()
    sig = three_compartment_fit((b[None, :], te[None, :]), *(p[:, k:k + 1] for k in range(6)), vol[:, 0:1], vol[:, 1:2]) / 1000.0
    return 1000.0 * (sig + rng.normal(0, noise, sig.shape))


def three_compartment_fit(M, D_ep, D_st, D_lu, T2_ep, T2_st, T2_lu, V_ep, V_st):
    """PIA.py:240-251 (host helper kept for the drop-in surface; the kernel evaluates the same expression)."""
    b, TE = M
    S_ep = V_ep * np.exp(-b / 1000 * D_ep) * np.exp(-TE / T2_ep)
    S_st = V_st * np.exp(-b / 1000 * D_st) * np.exp(-TE / T2_st)
    S_lu = (1 - V_ep - V_st) * np.exp(-b / 1000 * D_lu) * np.exp(-TE / T2_lu)
    return 1000 * (S_ep + S_st + S_lu)


def hybrid_fit_device(signals):
    """signals: [n, 16] (array or tensor, any float dtype) -> dict of device tensors
    params [n, 8] f64, status [n] i32 (scipy termination code), nfev [n] i32, cost [n] f64."""
    dev = ops.require_gpu()
    sig = signals if torch.is_tensor(signals) else torch.from_numpy(np.ascontiguousarray(np.asarray(signals, np.float64)))
    if sig.dim() != 2 or sig.shape[1] != 16:
        raise ValueError(f"signals must be [n_voxels, 16], got {tuple(sig.shape)}")
    sig = sig.to(dev, torch.float64).contiguous()
    if not bool(torch.isfinite(sig).all()):
        raise ValueError("array must not contain infs or NaNs")     # curve_fit(check_finite=True)
    n = sig.shape[0]
    params = torch.empty((n, 8), dtype=torch.float64, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    nfev = torch.empty(n, dtype=torch.int32, device=dev)
    cost = torch.empty(n, dtype=torch.float64, device=dev)
    if n:
        check(lib().inr_hybrid_fit(params.data_ptr(), status.data_ptr(), nfev.data_ptr(), cost.data_ptr(),
                                   sig.data_ptr(), n, ops._stream()), "inr_hybrid_fit")
    return {"params": params, "status": status, "nfev": nfev, "cost": cost}


def hybrid_fit(signals):
    """PIA.py:253-283: returns numpy D [n, 3], T2 [n, 3], v [n, 3] (v[:, 2] = 1 - V_ep - V_st)."""
    x = hybrid_fit_device(signals)["params"].cpu().numpy()
    n = x.shape[0]
    D, T2, v = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
    D[:, :] = x[:, 0:3]
    T2[:, :] = x[:, 3:6]
    v[:, 0:2] = x[:, 6:8]
    v[:, 2] = 1 - x[:, 6] - x[:, 7]
    return D, T2, v
