"""A/B of several debug-switch settings on the fused step: python tools/ab_keys.py "6=0,7=1" "6=1,7=0" ..."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import _lib, ops
from oracle.torch_port import fourier_matrix
lib = _lib.lib()
n = 64 * 64 * 128
x = ops.grid_fourier_map((64, 64, 128), torch.from_numpy(fourier_matrix(3)).cuda())
tgt = torch.rand(n, device="cuda")
for rep in range(2):
    for cfg in sys.argv[1:]:
        for kv in cfg.split(","):
            k, v = kv.split("=")
            lib.inr_debug_set(int(k), int(v))
        torch.manual_seed(0)
        net = inr.Siren(256, 512, 3, 1).cuda()
        fit = inr.SirenFitter(net)
        fit.step(x, tgt, 3)
        ops.prof_enable(True); ops.prof_reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fit.step(x, tgt, 10)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        res = [ops.prof_read(k) for k in range(4)]
        ops.prof_enable(False)
        print(f"{cfg:12s}: step {dt*1e3:.3f} ms | fwd {res[0][1]/res[0][0]:.3f}  dX {res[1][1]/res[1][0]:.3f}  dW {res[2][1]/res[2][0]:.3f}  other/step {res[3][1]/10:.3f}", flush=True)
