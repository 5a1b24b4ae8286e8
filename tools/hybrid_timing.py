"""Time the hybrid-fit kernel on slice-sized batches and the CPU oracle twin / scipy on a small sample."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from mri_super_resolution_amd import pia  # noqa: E402
from oracle import pia_oracle as P  # noqa: E402


def main():
    for n, noise in ((14400, 0.0), (14400, 0.02), (14400, 0.1), (57600, 0.02), (230400, 0.02)):
        sig = torch.from_numpy(P.synthetic_signals(n, noise, seed=5)).cuda()
        pia.hybrid_fit_device(sig)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = pia.hybrid_fit_device(sig)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        nf = out["nfev"].double()
        print(f"n={n} noise={noise}: {dt*1e3:.1f} ms  {n/dt:.0f} voxels/s  nfev mean {nf.mean():.1f} max {int(nf.max())} "
              f"status0 {(out['status']==0).sum().item()}", flush=True)
    from scipy.optimize import curve_fit
    sig = P.synthetic_signals(64, 0.02, seed=5)
    f = lambda M, *p: P.three_compartment(np.array(p), M[0], M[1])
    t0 = time.perf_counter()
    for y in sig:
        try:
            curve_fit(f, np.vstack([P.B16, P.TE16]), y, p0=list(P.P0), bounds=(list(P.LB), list(P.UB)), method="trf", maxfev=5000)
        except RuntimeError:
            pass
    dt = time.perf_counter() - t0
    print(f"scipy curve_fit on this host, 1 thread: {64/dt:.1f} voxels/s")


if __name__ == "__main__":
    main()
