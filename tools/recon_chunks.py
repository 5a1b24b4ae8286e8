#!/usr/bin/env python3
"""Dense x4 re-sampling of the synthetic 128^3 (256 x 256 x 128 = 8.4 M voxels) by chunk size: do chunks whose ping-pong activation
buffers fit the 256 MB Infinity Cache (32 k - 128 k rows) beat the 1 M-row default?  python tools/recon_chunks.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import mri_super_resolution_amd as inr  # noqa: E402
from mri_super_resolution_amd import drivers  # noqa: E402

torch.manual_seed(0)
net = inr.Siren(256, 512, 3, 1).cuda()
B = torch.from_numpy(drivers.fourier_matrix(3, seed=0)).cuda()
shape = (256, 256, 128)
for rep in range(2):
    for chunk in (1 << 15, 1 << 16, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21):
        rec = inr.reconstruct(net, shape, B, chunk_rows=chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            rec = inr.reconstruct(net, shape, B, chunk_rows=chunk)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"round {rep} chunk {chunk:8d} rows: {dt * 1e3:7.2f} ms, {rec.numel() / dt / 1e6:6.1f} M voxels/s", flush=True)
