import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import ops
N, H = 524288, 512
x = torch.randn(N, H, device='cuda'); W = torch.randn(H, H, device='cuda') * 0.05; b = torch.randn(H, device='cuda')
def timeit(f, reps=6):
    f(); f(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
fl = 2 * N * H * H / 1e9
for name, f in (("stash", lambda: ops.sine_layer_forward(x, W, b, 30.0, True)), ("nostash", lambda: ops.sine_layer_forward(x, W, b, 30.0, False)),
                ("plain dX", lambda: ops.sine_layer_backward_input(x, W, None))):
    t = timeit(f); print(f"{name}: {t:.3f} ms {fl/t:.1f} TF")
