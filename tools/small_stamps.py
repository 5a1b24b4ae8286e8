"""Phase timing inside the persistent small-net kernel (csrc/siren_small.hip), last step of a 64-step launch.
Needs the diagnostic build: python mri-super-resolution_amd/_build.py --diag -DINR_STAMPS, then
INR_LIB=$PWD/mri-super-resolution_amd/libinrhip_diag.so python tools/small_stamps.py [side]"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import _lib
lib = _lib.lib()
lib.inr_debug_set_ptr.argtypes = [ctypes.c_int, ctypes.c_void_p]
side = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = side * side
torch.manual_seed(0)
net = inr.Siren(2, 64, 6, 1).cuda()
coords = inr.ImageFitting_set([np.zeros((side, side), np.float32)]).coords[0]
tg = torch.rand(4, n, device="cuda") * 2 - 1
wt = torch.rand(4, n, device="cuda")
f = inr.SirenFitter(net, lr=3e-4)
f.step_cycle(coords, tg, 64, wt)
torch.cuda.synchronize()
nb = (n + 31) // 32
st = torch.zeros(nb * 8 * 16, dtype=torch.int64, device="cuda")
lib.inr_debug_set_ptr(0, st.data_ptr())
f.step_cycle(coords, tg, 64, wt)
torch.cuda.synchronize()
lib.inr_debug_set_ptr(0, None)
s = st.cpu().numpy().reshape(-1, 16).astype(np.float64)
names = ["stage W0/x + barrier", "forward (7 layers)", "head + dz + gW_head", "backward", "grid barrier 1", "reduce + Adam", "grid barrier 2"]
print(f"N={n}, {nb} blocks; ticks of s_memtime (100 MHz = 10 ns), medians over waves [p10 .. p90]")
for i, nm in enumerate(names):
    d = s[:, i + 1] - s[:, i]
    print(f"  {nm:24s} {np.median(d):7.0f}  [{np.percentile(d, 10):.0f} .. {np.percentile(d, 90):.0f}]")
print(f"  step total               {np.median(s[:, 7] - s[:, 0]):7.0f}")
