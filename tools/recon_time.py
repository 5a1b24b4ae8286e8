#!/usr/bin/env python3
"""Dense x4 re-sampling of the synthetic 128^3 (256 x 256 x 128 = 8.4 M voxels, Siren(256,512,3,1)): voxels/s with the
cross-layer fused forward (default) and with the layer-wise launches (inr_debug_set(19, 0))."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import mri_super_resolution_amd as inr  # noqa: E402
from mri_super_resolution_amd import drivers, ops  # noqa: E402

torch.manual_seed(0)
net = inr.Siren(256, 512, 3, 1).cuda()
B = torch.from_numpy(drivers.fourier_matrix(3, seed=0)).cuda()
shape = (256, 256, 128)
for key, name in ((1, "fused forward"), (0, "layer-wise")):
    with ops.debug_switch(19, key):
        rec = inr.reconstruct(net, shape, B)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            rec = inr.reconstruct(net, shape, B)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"{name}: {dt * 1e3:.2f} ms per re-sampling, {rec.numel() / dt / 1e6:.1f} M voxels/s, "
              f"{rec.numel() * 1836032 / dt / 1e12:.0f} TFLOP/s algorithmic")
