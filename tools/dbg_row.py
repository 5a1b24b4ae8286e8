"""GPU box: gemm_hp_row_kernel against the deferred-epilogue kernel, tensor by tensor (gradients of one step, forward output)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import ops
from mri_super_resolution_amd._lib import lib
fin, n = int(sys.argv[1]) if len(sys.argv) > 1 else 256, int(sys.argv[2]) if len(sys.argv) > 2 else 5197
g = torch.Generator().manual_seed(n)
x = (torch.rand(n, fin, generator=g) * 2 - 1).cuda()
t = torch.rand(n, generator=g).cuda()
res = {}
for row in (0, 1):
    lib().inr_debug_set(18, 0); lib().inr_debug_set(27, row); lib().inr_debug_set(28, 1)
    torch.manual_seed(0)
    net = inr.Siren(fin, 512, 2, 1).cuda()
    desc, flat = inr.flat_parameters(net)
    grads = torch.zeros_like(flat); loss = torch.zeros(1, device="cuda")
    ops.launch_counts_reset()
    ws = ops.siren_loss_grad(desc, flat, grads, x, t, None, 0, loss)
    c = ops.launch_counts()
    y = ops.siren_forward(desc, flat, x)
    res[row] = (grads.cpu().numpy(), float(loss), y.cpu().numpy(), {k: v for k, v in c.items() if v})
    lib().inr_debug_reset()
print(res[0][3], res[1][3])
print("loss", res[0][1], res[1][1], "y equal", np.array_equal(res[0][2], res[1][2]))
total, offs = ops.siren_param_layout(desc)
g0, g1 = res[0][0], res[1][0]
for l, (wo, bo) in enumerate(offs):
    wn = (bo - wo)
    a, b = g0[wo:bo], g1[wo:bo]
    nb = 512 if l < len(offs) - 1 else 1
    ab, bb = g0[bo:bo + nb], g1[bo:bo + nb]
    print(f"layer {l}: W differ {int((a != b).sum())} of {a.size} (max rel {np.abs(a - b).max() / (np.abs(a).max() + 1e-30):.2e}); "
          f"b differ {int((ab != bb).sum())} of {nb} (max rel {np.abs(ab - bb).max() / (np.abs(ab).max() + 1e-30):.2e})")
