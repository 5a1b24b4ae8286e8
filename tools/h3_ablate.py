"""Ablation timing of the split-fp16 GEMM kernels inside the fused step (results are garbage by construction)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import _lib, ops
from oracle.torch_port import fourier_matrix
lib = _lib.lib()
n = 64 * 64 * 128
x = ops.grid_fourier_map((64, 64, 128), torch.from_numpy(fourier_matrix(3)).cuda())
tgt = torch.rand(n, device="cuda")
for bits in [int(a) for a in sys.argv[1:]] or [0]:
    torch.manual_seed(0)
    net = inr.Siren(256, 512, 3, 1).cuda()
    fit = inr.SirenFitter(net)
    lib.inr_debug_set(4, bits)
    fit.step(x, tgt, 2)
    ops.prof_enable(True); ops.prof_reset()
    fit.step(x, tgt, 5)
    torch.cuda.synchronize()
    res = [ops.prof_read(k) for k in range(4)]
    ops.prof_enable(False)
    lib.inr_debug_set(4, 0)
    print(f"ablate={bits:2d}: fwd {res[0][1]/res[0][0]:.3f} ms  dX {res[1][1]/res[1][0]:.3f} ms  dW {res[2][1]/res[2][0]:.3f} ms  other/step {res[3][1]/5:.3f} ms", flush=True)
