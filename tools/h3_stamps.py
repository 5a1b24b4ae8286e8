"""Per-wave s_memtime stamps of the split-fp16 GEMM (needs tools/build_stamps.sh; stand-alone debug mode of the layer
calls).  Prints where a wave's lifetime goes and, for K-tile 4, the duration of each MFMA block."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import ops, _lib
lib = _lib.lib()
lib.inr_debug_set_ptr.argtypes = [ctypes.c_int, ctypes.c_void_p]
scratch = torch.zeros(32 << 20, dtype=torch.uint8, device="cuda")
lib.inr_debug_set_ptr(1, scratch.data_ptr())
lib.inr_debug_set(3, 2)
lib.inr_debug_set(6, 0)      # 128 x 128 tiles everywhere: 4 waves per block
N, H = 524288, 512
x = torch.rand(N, H, device="cuda") * 2 - 1
W = (torch.rand(H, H, device="cuda") * 2 - 1) * 0.0036
b = torch.randn(H, device="cuda") * 0.01
dz = torch.randn(N, H, device="cuda") * 1e-7
dact = torch.randn(N, H, device="cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
nwaves = {"fwd": N // 128 * 4 * 4, "dx": N // 128 * 4 * 4, "dw": 16 * 64 * 4}[which]
st = torch.zeros(nwaves * 16, dtype=torch.int64, device="cuda")


def run():
    if which == "fwd":
        ops.sine_layer_forward(x, W, b, 30.0, True)
    elif which == "dx":
        ops.sine_layer_backward_input(dz, W, dact)
    else:
        ops.linear_param_grad(dz, x, False)


run(); torch.cuda.synchronize()
lib.inr_debug_set_ptr(0, st.data_ptr()); run(); torch.cuda.synchronize(); lib.inr_debug_set_ptr(0, None)
s = st.cpu().numpy().reshape(-1, 16).astype(np.float64)
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
med = lambda a: float(np.median(a))
print(f"{which}: {len(s)} waves, kernel span {(s[:, 9].max() - t0):.0f} ticks (s_memtime)")
print("per-wave medians: prologue %.0f | K-loop %.0f | park acc %.0f | epilogue rows %.0f | total %.0f" % (
    med(s[:, 1] - s[:, 0]), med(s[:, 7] - s[:, 1]), med(s[:, 8] - s[:, 7]), med(s[:, 9] - s[:, 8]), med(s[:, 9] - s[:, 0])))
ok = s[:, 2] > 0
print("K-tile 4: X block (split+park+16 MFMA) %.0f | A loads + Y block %.0f | barrier %.0f | B loads + Z + frag reads %.0f | total %.0f" % (
    med(s[ok, 3] - s[ok, 2]), med(s[ok, 4] - s[ok, 3]), med(s[ok, 5] - s[ok, 4]), med(s[ok, 6] - s[ok, 5]), med(s[ok, 6] - s[ok, 2])))
for q in (10, 50, 90):
    print(f"   p{q}: X {np.percentile(s[ok,3]-s[ok,2], q):.0f}  Y {np.percentile(s[ok,4]-s[ok,3], q):.0f}  barrier {np.percentile(s[ok,5]-s[ok,4], q):.0f}  Z {np.percentile(s[ok,6]-s[ok,5], q):.0f}")
lib.inr_debug_set(3, 1); lib.inr_debug_set(6, 1)
