"""Timing of the conv3d building blocks of the RAMS training step at the network's working size (batch 25)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import rams
B, D1, D2, D3 = 25, 130, 130, 9
x = torch.randn(B, D1, D2, D3, 32, device="cuda")
dy = torch.randn(B, D1, D2, D3, 32, device="cuda")
w = torch.randn(27, 32, 32, device="cuda") * 0.05
b = torch.zeros(32, device="cuda")
fl = 2.0 * B * D1 * D2 * D3 * 27 * 32 * 32


def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for name, f in (("forward", lambda: rams.conv3d(x, w, b)), ("dgrad", lambda: rams.conv3d_dgrad(dy, w)),
                ("wgrad", lambda: rams.conv3d_wgrad(x, dy))):
    dt = t(f)
    print(f"{name:8s} {dt*1e3:7.3f} ms  {fl/dt/1e12:6.1f} TFLOP/s", flush=True)
