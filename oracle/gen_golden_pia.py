#!/usr/bin/env python3
"""Generate tests/golden/pia_hybrid.npz by running the REAL reference `PIA.hybrid_fit` (build container only).

Run from the repo root:  python oracle/gen_golden_pia.py
Inputs: seeded synthetic three-compartment signals (oracle/pia_oracle.synthetic_signals) at four noise levels.
Outputs: the reference's D, T2, v (PIA.py:253-283) plus, per voxel, the nfev / cost / status that scipy's curve_fit
reports for the reference's own model function with the reference's arguments.  Only data is written.
"""
import os
import sys
import warnings

import numpy as np
from scipy.optimize import curve_fit

warnings.filterwarnings("ignore")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/implicit-neural-representations")
import PIA  # noqa: E402  (the reference module itself)
import pia_oracle as P  # noqa: E402


def main():
    parts, noise = [], []
    for k, nz in enumerate((0.0, 0.005, 0.02, 0.1)):
        parts.append(P.synthetic_signals(32, nz, seed=100 + k))
        noise += [nz] * 32
    signals = np.concatenate(parts)
    D, T2, v = PIA.hybrid_fit(signals)
    X, Y = np.meshgrid([0, 13, 93, 143], [0, 150, 1000, 1500])
    xdata = np.vstack((Y.ravel(), X.ravel()))
    nfev, cost, status = [], [], []
    for y in signals:
        try:
            _, _, info, _, ier = curve_fit(PIA.three_compartment_fit, xdata, y, p0=[0.55, 1.3, 2.8, 50, 70, 750, 0.3, 0.4],
                                           check_finite=True, bounds=([0.3, 0.7, 2.7, 20, 40, 500, 0, 0],
                                                                      [0.7, 1.7, 3.0, 70, 100, 1000, 1, 1]),
                                           method="trf", maxfev=5000, full_output=True)
            nfev.append(info["nfev"]); cost.append(0.5 * float(info["fvec"] @ info["fvec"])); status.append(ier)
        except RuntimeError:
            nfev.append(5000); cost.append(np.nan); status.append(0)
    out = os.path.join(HERE, "..", "tests", "golden", "pia_hybrid.npz")
    np.savez_compressed(out, signals=signals, noise=np.array(noise), D=D, T2=T2, v=v, nfev=np.array(nfev),
                        cost=np.array(cost), status=np.array(status))
    print("pia_hybrid.npz", os.path.getsize(out), "bytes; status counts", np.bincount(status))


if __name__ == "__main__":
    main()
