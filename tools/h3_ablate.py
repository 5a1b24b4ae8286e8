"""Timing of the split-fp16 GEMM kernels inside the fused step.  For the ablation table of DESIGN.md build the library
with -DH3_ABLATE=<bits> (1 no K-loop loads, 2 no split/park, 4 no fragment reads, 8 no epilogue; results are garbage by
construction) and pass the same value as a label."""
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
    pass  # needs a -DH3_ABLATE=bits build (see gemm_h3.inc)
    fit.step(x, tgt, 2)
    ops.prof_enable(True); ops.prof_reset()
    fit.step(x, tgt, 5)
    torch.cuda.synchronize()
    res = [ops.prof_read(k) for k in range(4)]
    ops.prof_enable(False)
    pass
    print(f"ablate={bits:2d}: fwd {res[0][1]/res[0][0]:.3f} ms  dX {res[1][1]/res[1][0]:.3f} ms  dW {res[2][1]/res[2][0]:.3f} ms  other/step {res[3][1]/5:.3f} ms", flush=True)
