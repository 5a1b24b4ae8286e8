#!/usr/bin/env python3
"""`python tools/fit_n.py ROWS [STEPS]`: STEPS fused fit steps of Siren(256,512,3,1) on ROWS synthetic rows (for rocprofv3
kernel traces of one row count); prints ms per step."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import mri_super_resolution_amd as inr  # noqa: E402

n = int(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
g = torch.Generator(device="cuda").manual_seed(n)
x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous()
t = torch.rand(n, device="cuda", generator=g)
torch.manual_seed(0)
f = inr.SirenFitter(inr.Siren(256, 512, 3, 1).cuda(), lr=1e-4)
f.step(x, t, 5)
torch.cuda.synchronize()
t0 = time.perf_counter()
f.step(x, t, steps)
torch.cuda.synchronize()
print(f"rows {n}: {(time.perf_counter() - t0) / steps * 1e3:.4f} ms per step over {steps} steps")
