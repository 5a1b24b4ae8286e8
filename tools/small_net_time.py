"""Small-net regime timing (master.py: Siren(2, 64, 6, 1) on a 60x60 slice): microseconds per optimizer step, one C call for
all steps and one call per step.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 60
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
torch.manual_seed(0)
net = inr.Siren(2, 64, 6, 1).cuda()
coords = inr.ImageFitting_set([np.zeros((side, side), np.float32)]).coords[0]
tgt = torch.rand(side * side, 1, device="cuda") * 2 - 1
w = torch.rand(side * side, 1, device="cuda")
f = inr.SirenFitter(net, lr=3e-4)
f.step(coords, tgt, 10, w)
torch.cuda.synchronize()
t0 = time.perf_counter()
f.step(coords, tgt, steps, w)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(steps):
    f.step(coords, tgt, 1, w)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"N={side * side}: one call {1e6 * (t1 - t0) / steps:.1f} us/step, call per step {1e6 * (t2 - t1) / steps:.1f} us/step")
