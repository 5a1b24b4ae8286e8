#!/usr/bin/env python3
"""Which seed ends on an Adam spike at step 2,500 is decided by the last bits: the three seeds that do so in profiles/r05_t4_paths.json
(14 on the product path, 13 on the exact-fp32 kernels, 3 on round 1's split kernels) re-run with the SAME arithmetic and another
summation order of the parameter gradients (debug key 20 = 0: one launch per layer instead of the merged one; key 22 = 128: other row
splits of the merged launch) or another MFMA shape of the exact-fp32 kernels (key 1 = 0: 32x32x2 instead of 16x16x4).
    python tools/t4_seed_chaos.py [seed ...]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import drivers, metrics
from mri_super_resolution_amd._lib import check, lib

ARMS = (("split_fp16 (product)", ()), ("split_fp16, param-grads one launch per layer", ((20, 0),)), ("split_fp16, merged launch on 128 blocks", ((22, 128),)),
        ("exact_fp32 (16x16x4)", ((3, 0),)), ("exact_fp32 (32x32x2)", ((3, 0), (1, 0))))
z = np.load(os.path.join(ROOT, "tests", "golden", "pat07_slice11.npz"))
hr, lr = np.ascontiguousarray(z["hr"], np.float32), np.ascontiguousarray(z["lr"], np.float32)
hr_t = torch.from_numpy(hr).cuda()
seeds = [int(a) for a in sys.argv[1:]] or [14, 13, 3, 0]
print("PSNR (dB) at step 2,500 [max loss over the last 500 steps]; config 1, pat07 slice 11")
for name, keys in ARMS:
    lib().inr_debug_reset()
    for k, v in keys:
        check(lib().inr_debug_set(k, v), "inr_debug_set")
    cells = []
    for s in seeds:
        torch.manual_seed(s)
        B = torch.from_numpy(drivers.fourier_matrix(2, seed=s)).cuda()
        net = inr.Siren(256, 512, 3, 1).cuda()
        ds = inr.ImageFitting_set([lr])
        x = inr.input_mapping(ds.coords[0], B)
        f = inr.SirenFitter(net, lr=1e-4)
        losses = f.step(x, ds.pixels[0], 2500).cpu().numpy()
        p = float(metrics.psnr(hr_t, inr.reconstruct(net, (128, 128), B), 1.0))
        cells.append(f"seed {s:2d}: {p:6.2f} [{losses[-500:].max():.1e}]")
    print(f"{name:48s} " + "   ".join(cells), flush=True)
lib().inr_debug_reset()
