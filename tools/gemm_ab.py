"""A/B of the two pipelined GEMM kernels (32x32x2 vs 16x16x4 MFMA) in one process + equality check."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import ops, _lib
lib = _lib.lib()
N, H = 524288, 512
x = torch.randn(N, H, device='cuda'); W = torch.randn(H, H, device='cuda') * 0.05; b = torch.randn(H, device='cuda')
dz = torch.randn(N, H, device='cuda'); dact = torch.randn(N, H, device='cuda')
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
fl = 2 * N * H * H / 1e9
res = {}
for rnd in range(3):
    for mode in (0, 1):
        lib.inr_debug_set(1, mode)
        t1 = timeit(lambda: ops.sine_layer_forward(x, W, b, 30.0, True))
        t2 = timeit(lambda: ops.sine_layer_backward_input(dz, W, dact))
        t3 = timeit(lambda: ops.linear_param_grad(dz, x, False))
        print(f"round {rnd} mfma16={mode}: fwd {fl/t1:.1f} | dX {fl/t2:.1f} | dW {fl/t3:.1f} TF", flush=True)
outs = {}
for mode in (0, 1):
    lib.inr_debug_set(1, mode)
    a, d = ops.sine_layer_forward(x[:4096], W, b, 30.0, True)
    g = ops.sine_layer_backward_input(dz[:4096], W, dact[:4096])
    gw, gb = ops.linear_param_grad(dz[:8192], x[:8192], True)
    outs[mode] = [t.double() for t in (a, d, g, gw, gb)]
lib.inr_debug_set(1, 0)
for name, u, v in zip(("act", "dact", "dx", "gW", "gb"), outs[0], outs[1]):
    print(name, "rel diff 16x16x4 vs 32x32x2:", ((u - v).norm() / u.norm()).item())
ref = (dz[:8192].double().T @ x[:8192].double())
print("gW vs fp64:", ((outs[1][3] - ref).norm() / ref.norm()).item(), ((outs[0][3] - ref).norm() / ref.norm()).item())
