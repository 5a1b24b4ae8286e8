"""GPU box: RAMS training step, debug key 24 (epilogue-fused ReLU mask / residual sum in the data-gradient convolutions) on and off,
interleaved: ms per step at batch 32 of 32x32x9."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import rams
from mri_super_resolution_amd._lib import lib
B, P = 32, 32
rng = np.random.default_rng(0)
tr = rams.RamsTrainer(rams.RAMS(seed=0))
lr = (rng.random((B, P, P, 9)) * 20000).astype(np.float32)
hr = (rng.random((B, 3 * P, 3 * P, 1)) * 20000).astype(np.float32)
mask = np.ones((B, 3 * P, 3 * P, 1), np.float32)
fwd = 265e9 * (P * P) / (128 * 128) * B
for rnd in range(3):
    for key in (1, 0):
        lib().inr_debug_set(24, key)
        tr.train_step(lr, hr, mask); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            tr.train_step(lr, hr, mask)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"round {rnd} key24={key}: {dt * 1e3:.3f} ms per step, {3 * fwd / dt / 1e12:.1f} TFLOP/s", flush=True)
lib().inr_debug_set(24, 2)
