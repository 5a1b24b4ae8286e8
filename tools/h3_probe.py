"""Split-fp16 GEMM probe: accuracy against fp64 and speed against the fp32-MFMA kernels, through the standalone layer
entry points (debug mode inr_debug_set(3, 2): planes/amax built per call in a scratch buffer)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import _lib, ops  # noqa: E402

lib = _lib.lib()
lib.inr_debug_set_ptr.argtypes = [ctypes.c_int, ctypes.c_void_p]
scratch = torch.zeros(32 << 20, dtype=torch.uint8, device="cuda")
lib.inr_debug_set_ptr(1, scratch.data_ptr())
N, H = int(sys.argv[1]) if len(sys.argv) > 1 else 524288, 512
torch.manual_seed(0)
x = torch.rand(N, H, device="cuda") * 2 - 1
W = (torch.rand(H, H, device="cuda") * 2 - 1) * 0.0036
b = torch.randn(H, device="cuda") * 0.01
dz = torch.randn(N, H, device="cuda") * torch.exp(torch.randn(N, 1, device="cuda") * 2) * 1e-7
dact = torch.randn(N, H, device="cuda")


def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


rel = lambda a, r: ((a.double() - r).norm() / r.norm()).item()
rows = torch.arange(0, N, max(1, N // 2048), device="cuda")
ref_f = torch.sin(30 * (x[rows].double() @ W.double().T + b.double()))
ref_dx = (dz[rows].double() @ W.double()) * dact[rows].double()
ref_dw = dz.double().T @ x.double()
fl = 2 * N * H * H / 1e9
for mode in (0, 2):
    lib.inr_debug_set(3, mode)
    a, d = ops.sine_layer_forward(x, W, b, 30.0, True)
    dx = ops.sine_layer_backward_input(dz, W, dact)
    gW, _ = ops.linear_param_grad(dz, x, False)
    print(f"mode {mode}: err fwd {rel(a[rows], ref_f):.2e}  dX {rel(dx[rows], ref_dx):.2e}  dW {rel(gW, ref_dw):.2e}", flush=True)
    ops.prof_enable(True); ops.prof_reset()
    t1 = timeit(lambda: ops.sine_layer_forward(x, W, b, 30.0, True))
    t2 = timeit(lambda: ops.sine_layer_backward_input(dz, W, dact))
    t3 = timeit(lambda: ops.linear_param_grad(dz, x, False))
    print(f"mode {mode}: wall fwd {t1:.3f} ms {fl/t1:.0f} TF | dX {t2:.3f} ms {fl/t2:.0f} TF | dW {t3:.3f} ms {fl/t3:.0f} TF", flush=True)
    ops.prof_enable(False)
lib.inr_debug_set(3, 0)
