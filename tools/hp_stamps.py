"""Per-wave s_memtime stamps of the pre-split GEMM (gemm_hp.inc) inside a fused fit step.
Needs the diagnostic build: python mri-super-resolution_amd/_build.py --diag -DINR_STAMPS, then
INR_LIB=$PWD/mri-super-resolution_amd/libinrhip_diag.so python tools/hp_stamps.py [fwd|dx|dw] [nth]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import _lib
lib = _lib.lib()
lib.inr_debug_set_ptr.argtypes = [ctypes.c_int, ctypes.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = 524288
torch.manual_seed(0)
net = inr.Siren(256, 512, 3, 1).cuda()
x = torch.rand(N, 256, device="cuda") * 2 - 1
t = torch.rand(N, 1, device="cuda")
fitter = inr.SirenFitter(net, lr=1e-4)
fitter.step(x, t, 3)
torch.cuda.synchronize()
nblocks = {"fwd": N // 128 * 2, "dx": N // 128 * 2, "dw": 4096}[which]
st = torch.zeros(nblocks * 8 * 16, dtype=torch.int64, device="cuda")
lib.inr_debug_set(8, {"fwd": 0, "dx": 1, "dw": 2}[which])
lib.inr_debug_set(9, nth)
lib.inr_debug_set_ptr(0, st.data_ptr())
fitter.step(x, t, 1)
torch.cuda.synchronize()
lib.inr_debug_set_ptr(0, None)
s = st.cpu().numpy().reshape(-1, 16).astype(np.float64)
s = s[s[:, 0] > 0]
med = lambda a: float(np.median(a))
print(f"{which} launch #{nth}: {len(s)} waves ({len(s) // 8} blocks), kernel span {s[:, 4].max() - s[:, 0].min():.0f} ticks of s_memtime (100 MHz)")
print("per-wave medians [ticks]: prologue %.0f | K-loop %.0f | mul loads + barrier + park %.0f | epilogue rows %.0f | total %.0f" % (
    med(s[:, 1] - s[:, 0]), med(s[:, 2] - s[:, 1]), med(s[:, 3] - s[:, 2]), med(s[:, 4] - s[:, 3]), med(s[:, 4] - s[:, 0])))
ok = s[:, 5] > 0
if ok.any():
    print("K-tile 7: X %.0f | Y %.0f | wait + barrier %.0f | Z (+ DMA issue, fragment reads) %.0f | total %.0f" % (
        med(s[ok, 6] - s[ok, 5]), med(s[ok, 7] - s[ok, 6]), med(s[ok, 8] - s[ok, 7]), med(s[ok, 9] - s[ok, 8]), med(s[ok, 9] - s[ok, 5])))
    for q in (10, 50, 90):
        print(f"   p{q}: X {np.percentile(s[ok,6]-s[ok,5], q):.0f}  Y {np.percentile(s[ok,7]-s[ok,6], q):.0f}  barrier {np.percentile(s[ok,8]-s[ok,7], q):.0f}  Z {np.percentile(s[ok,9]-s[ok,8], q):.0f}")
# blocks per CU over time: how many tiles does a CU run back to back, and how long is the gap between them

# deferred-epilogue kernel (gemm_hp_pkd): iterations 6 (waves 0-3 convert a chunk) and 7 (waves 4-7 do), per wave half
w = np.arange(len(s)) % 8
for name, base in (("iteration 6", 5), ("iteration 7", 10)):
    okk = s[:, base] > 0
    if not okk.any():
        continue
    for half, sel in (("waves 0-3", okk & (w < 4)), ("waves 4-7", okk & (w >= 4))):
        d = [med(s[sel, base + k + 1] - s[sel, base + k]) for k in range(4)]
        print(f"{name} {half}: X {d[0]:.0f} | wait+barrier {d[1]:.0f} | chunk {d[2]:.0f} | Z+Y {d[3]:.0f} | total {sum(d):.0f}")
