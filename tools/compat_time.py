"""GPU box: ms per step of the reference's loop verbatim (autograd + torch.optim.Adam) on the fused kernels, on the layer-by-layer
path and of the fused fit entry point, at a given row count:  python tools/compat_time.py [rows ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mri_super_resolution_amd as inr  # noqa: E402
from mri_super_resolution_amd import inr as inr_mod  # noqa: E402


def loop(x, t, hp, k):
    inr_mod.HP_AUTOGRAD = hp
    torch.manual_seed(0)
    INR = inr.Siren(256, 512, 3, 1).cuda()
    opt = torch.optim.Adam(lr=1e-4, params=INR.parameters())
    for ctr in range(3 + k):
        if ctr == 3:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = INR(x)
        loss = ((out - t) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    inr_mod.HP_AUTOGRAD = True
    return (time.perf_counter() - t0) / k * 1e3, float(loss)


for n in [int(a) for a in sys.argv[1:]] or [524288, 114688, 4096]:
    x = torch.rand(n, 256, device="cuda") * 2 - 1
    t = torch.rand(n, 1, device="cuda")
    k = 20 if n > 100000 else 200
    a, la = loop(x, t, True, k)
    b, lb = loop(x, t, False, max(5, k // 2))
    torch.manual_seed(0)
    net = inr.Siren(256, 512, 3, 1).cuda()
    f = inr.SirenFitter(net, lr=1e-4)
    f.step(x, t, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = f.step(x, t, k)
    torch.cuda.synchronize()
    c = (time.perf_counter() - t0) / k * 1e3
    print(f"rows {n}: verbatim loop on the fused kernels {a:.3f} ms/step (loss {la:.3e}), layer by layer {b:.3f} (loss {lb:.3e}), "
          f"fused fit {c:.3f} (loss {float(losses[-1]):.3e}); ratio {a / c:.2f}", flush=True)
    del f, net
    torch.cuda.empty_cache()
