import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import ops, _lib
lib = _lib.lib()
torch.manual_seed(0)
n, fin, fout = 256, 64, 128
x = torch.randn(n, fin, device='cuda'); W = torch.randn(fout, fin, device='cuda') * 0.02; b = torch.randn(fout, device='cuda') * 0.1
lib.inr_debug_set(0, 1); a0, d0 = ops.sine_layer_forward(x, W, b, 30.0, True)
lib.inr_debug_set(0, 0); a1, d1 = ops.sine_layer_forward(x, W, b, 30.0, True)
torch.cuda.synchronize()
diff = (a0 - a1).abs()
print("max diff act", diff.max().item(), "dact", (d0 - d1).abs().max().item())
bad = (diff > 1e-5).nonzero()
print("n bad", bad.shape[0], "of", n * fout)
if bad.shape[0]:
    rows = bad[:, 0].unique(); cols = bad[:, 1].unique()
    print("bad rows", rows[:40].tolist(), "...", rows.shape[0]); print("bad cols", cols[:40].tolist(), "...", cols.shape[0])
    # is a1 a permutation of a0 rows/cols?
    r0 = bad[0].tolist(); v = a1[r0[0], r0[1]].item()
    m = ((a0 - v).abs() < 1e-6).nonzero()
    print("fast value at", r0, "=", v, "found in generic at", m[:5].tolist())
print("--- no stash")
lib.inr_debug_set(0, 0); a2, _ = ops.sine_layer_forward(x, W, b, 30.0, False)
d = (a0 - a2).abs(); bad = (d > 1e-5).nonzero(); print("n bad", bad.shape[0], "cols", bad[:, 1].unique()[:16].tolist() if bad.shape[0] else [])
print("--- plain input grad")
dz = torch.randn(n, fout, device='cuda')
g = ops.sine_layer_backward_input(dz, W, None)
ref = dz.double() @ W.double()
d = (g.double() - ref).abs(); bad = (d > 1e-4).nonzero(); print("n bad", bad.shape[0], "cols", bad[:, 1].unique()[:16].tolist() if bad.shape[0] else [], "rows", bad[:, 0].unique()[:8].tolist() if bad.shape[0] else [])
mulm = torch.randn(n, fin, device='cuda')
g2 = ops.sine_layer_backward_input(dz, W, mulm)
d = (g2.double() - ref * mulm.double()).abs(); bad = (d > 1e-4).nonzero(); print("mul: n bad", bad.shape[0], "cols", bad[:, 1].unique()[:16].tolist() if bad.shape[0] else [])
