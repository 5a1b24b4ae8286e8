"""RAMS training step (utils/training.py:193-209) on the device: milliseconds per step at the reference's training shape
(batch 32 of 32x32x9 low-resolution patches -> 96x96 targets)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import rams
B, P = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 32
rng = np.random.default_rng(0)
model = rams.RAMS(seed=0)
tr = rams.RamsTrainer(model)
lr = (rng.random((B, P, P, 9)) * 20000).astype(np.float32)
hr = (rng.random((B, 3 * P, 3 * P, 1)) * 20000).astype(np.float32)
mask = np.ones((B, 3 * P, 3 * P, 1), np.float32)
tr.train_step(lr, hr, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    loss = tr.train_step(lr, hr, mask)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
# forward 265 GFLOP per 128x128 stack scales with the patch area; a step is ~3 forward-equivalents (forward, data grad, weight grad)
fwd = 265e9 * (P * P) / (128 * 128) * B
print(f"RAMS train_step batch {B} of {P}x{P}x9: {dt * 1e3:.1f} ms per step, ~{3 * fwd / dt / 1e12:.0f} TFLOP/s (32 -> 32 convolutions: forward, data and weight gradient on split-fp16 MFMA)")
