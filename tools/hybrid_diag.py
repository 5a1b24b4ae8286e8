import sys
import numpy as np
sys.path.insert(0, ".")
from mri_super_resolution_amd import pia
from mri_super_resolution_amd._lib import lib
from oracle import pia_oracle as P
sig = P.synthetic_signals(200, 0.02, seed=11)
a = pia.hybrid_fit_device(sig)
lib().inr_debug_set(2, 0)
b = pia.hybrid_fit_device(sig)
lib().inr_debug_set(2, 1)
xa, xb = a["params"].cpu().numpy(), b["params"].cpu().numpy()
ca = np.stack([P.three_compartment(p) for p in xa]); cb = np.stack([P.three_compartment(p) for p in xb])
err = np.linalg.norm(ca - cb, axis=1) / np.linalg.norm(cb, axis=1)
for i in np.argsort(-err)[:6]:
    print(i, err[i], "status", a["status"][i].item(), b["status"][i].item(), "nfev", a["nfev"][i].item(), b["nfev"][i].item(),
          "cost", a["cost"][i].item(), b["cost"][i].item())
    _, info = P.trf_fit(sig[i], True)
    print("   oracle:", info)
