"""Do the input-grad and param-grad GEMMs of one layer overlap when launched on two streams?  (split-fp16 kernels,
stand-alone debug mode; timing only)"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import _lib, ops
lib = _lib.lib()
lib.inr_debug_set_ptr.argtypes = [ctypes.c_int, ctypes.c_void_p]
scratch = torch.zeros(32 << 20, dtype=torch.uint8, device="cuda")
lib.inr_debug_set_ptr(1, scratch.data_ptr())
lib.inr_debug_set(3, 2)
N, H = 524288, 512
x = torch.rand(N, H, device="cuda") * 2 - 1
W = (torch.rand(H, H, device="cuda") * 2 - 1) * 0.0036
dz = torch.randn(N, H, device="cuda") * 1e-7
dact = torch.randn(N, H, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
out_dx = torch.empty(N, H, device="cuda")


def run(concurrent, reps=6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if concurrent:
            with torch.cuda.stream(s1):
                ops.sine_layer_backward_input(dz, W, dact, out=out_dx)
            with torch.cuda.stream(s2):
                ops.linear_param_grad(dz, x, False)
        else:
            ops.sine_layer_backward_input(dz, W, dact, out=out_dx)
            ops.linear_param_grad(dz, x, False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


run(False, 2); run(True, 2)
for _ in range(2):
    print(f"sequential {run(False):.3f} ms   two streams {run(True):.3f} ms", flush=True)
lib.inr_debug_set(3, 1)
