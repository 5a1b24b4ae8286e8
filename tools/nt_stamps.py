"""Anatomy of ONE narrow-tile GEMM launch of a small fit step (s_memrealtime stamps, 10 ns ticks, one counter for the chip):
when does every wave enter, see its first operands, leave the K-loop, finish its stores -- relative to the first wave's entry.
Needs the diagnostic build: python mri-super-resolution_amd/_build.py --diag -DINR_STAMPS, then
INR_LIB=$PWD/mri-super-resolution_amd/libinrhip_diag.so python tools/nt_stamps.py [rows] [fwd|dx] [nth]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import _lib
lib = _lib.lib()
lib.inr_debug_set_ptr.argtypes = [ctypes.c_int, ctypes.c_void_p]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
nth = int(sys.argv[3]) if len(sys.argv) > 3 else 1
torch.manual_seed(0)
net = inr.Siren(256, 512, 3, 1).cuda()
x = torch.rand(N, 256, device="cuda") * 2 - 1
t = torch.rand(N, 1, device="cuda")
fitter = inr.SirenFitter(net, lr=1e-4)
fitter.step(x, t, 20)
torch.cuda.synchronize()
nblocks = (N + 63) // 64 * 4
st = torch.zeros(nblocks * 8 * 16, dtype=torch.int64, device="cuda")
lib.inr_debug_set(8, {"fwd": 0, "dx": 1}[which])
lib.inr_debug_set(9, nth)
lib.inr_debug_set_ptr(0, st.data_ptr())
fitter.step(x, t, 1)
torch.cuda.synchronize()
lib.inr_debug_set_ptr(0, None)
s = st.cpu().numpy().reshape(-1, 16).astype(np.float64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
us = lambda a: (a - t0) / 100.0
print(f"rows {N}, {which} launch #{nth}: {len(s)} waves stamped ({len(s) // 4} blocks); everything in us after the first wave's entry")
for k, name in enumerate(("wave enters", "first K-tile visible (prologue done)", "K-loop done", "epilogue stores issued (wave ends)")):
    v = us(s[:, k])
    print(f"  {name:42s} min {v.min():6.2f}  p10 {np.percentile(v, 10):6.2f}  median {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f}")
d = lambda a, b: (s[:, b] - s[:, a]) / 100.0
print("  per-wave durations (median / p90): prologue %.2f / %.2f   K-loop %.2f / %.2f   epilogue %.2f / %.2f" % (
    np.median(d(0, 1)), np.percentile(d(0, 1), 90), np.median(d(1, 2)), np.percentile(d(1, 2), 90), np.median(d(2, 3)), np.percentile(d(2, 3), 90)))
