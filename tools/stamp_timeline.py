"""Diagnostic: per-wave s_memtime stamps of the pipelined GEMM (needs a -DINR_STAMPS build of the library,
see tools/build_stamps.sh).  Prints where a wave's lifetime goes and how busy each SIMD's two wave slots are."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("INR_LIB", "")
from mri_super_resolution_amd import ops, _lib
lib = _lib.lib()
N, H = 524288, 512
x = torch.randn(N, H, device='cuda'); W = torch.randn(H, H, device='cuda') * 0.05; b = torch.randn(H, device='cuda')
dz = torch.randn(N, H, device='cuda'); dact = torch.randn(N, H, device='cuda')
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
nblocks = (N // 128) * 4
st = torch.zeros(nblocks * 4 * 8, dtype=torch.int64, device='cuda')
def run():
    if which == "fwd": ops.sine_layer_forward(x, W, b, 30.0, True)
    elif which == "dx": ops.sine_layer_backward_input(dz, W, dact)
    elif which == "plain": ops.sine_layer_backward_input(dz, W, None)
run(); torch.cuda.synchronize()
lib.inr_debug_set_ptr(0, st.data_ptr()); run(); torch.cuda.synchronize(); lib.inr_debug_set_ptr(0, None)
s = st.cpu().numpy().reshape(-1, 8)
t0 = s[:, 0].min()
start, ml0, ml1, ep0, ep1 = [(s[:, i] - t0).astype(np.float64) for i in range(5)]
print("kernel span (cycles @100MHz ticks?)", ep1.max())
print("per-wave medians: prologue %.0f  mainloop %.0f  barrier %.0f  epilogue %.0f  total %.0f" % (
    np.median(ml0 - start), np.median(ml1 - ml0), np.median(ep0 - ml1), np.median(ep1 - ep0), np.median(ep1 - start)))
hw = s[:, 7]; xcc = (hw >> 32) & 15; hwid = hw & 0xffffffff
key = (xcc << 20) | (((hwid >> 13) & 7) << 16) | (((hwid >> 8) & 15) << 8) | (((hwid >> 4) & 3))
# for one SIMD: list intervals
k0 = key[0]; idx = np.where(key == k0)[0]; idx = idx[np.argsort(start[idx])]
print("epilogue split medians: stage-to-LDS %.0f  first 8 rows-groups %.0f  last 8 %.0f" % (np.median(s[:,5]-s[:,3]), np.median(s[:,6]-s[:,5]), np.median(s[:,4]-s[:,6])))
print("one SIMD, first 12 waves: start, ml0, ml1, ep0, ep1 (kcycles)")
for i in idx[:12]:
    print("  slot", int(hwid[i] & 15), ["%.1f" % (v / 1000) for v in (start[i], ml0[i], ml1[i], ep0[i], ep1[i])])
# overlap statistics: fraction of time exactly 0/1/2 waves are inside their MAIN LOOP on this SIMD
ev = []
for i in idx: ev += [(ml0[i], 1), (ml1[i], -1)]
ev.sort(); cur = 0; last = ev[0][0]; acc = {0: 0.0, 1: 0.0, 2: 0.0, 3: 0.0}
for t, d in ev:
    acc[min(cur, 3)] += t - last; last = t; cur += d
tot = sum(acc.values()); print("time with k waves in main loop on this SIMD:", {k: round(v / tot, 3) for k, v in acc.items()}, "span", tot)
