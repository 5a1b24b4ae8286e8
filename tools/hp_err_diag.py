#!/usr/bin/env python3
"""Per-tensor gradient error of the HL32 path and of the exact-fp32 kernels against float64 for one adversarial case of
tests/test_gpu_hp_variants.py (head weights x100), with the z-only stash on and off."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch  # noqa: E402

from mri_super_resolution_amd import ops  # noqa: E402
from oracle import inr_oracle as O  # noqa: E402
import test_gpu_hp_variants as T  # noqa: E402

import numpy as np  # noqa: E402

import mri_super_resolution_amd as inr  # noqa: E402
from oracle import torch_port as P  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "late":
    # the weights after the full 2,500-step config-1 fit (tests/test_gpu_hp_variants.py::test_late_training_state)
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    d, lr_img = np.load(os.path.join(G, "dataset_ff.npz")), np.load(os.path.join(G, "pat07_slice11.npz"))["lr"]
    torch.manual_seed(0)
    fitted = inr.Siren(256, 512, 3, 1).cuda()
    xd = inr.input_mapping(inr.get_mgrid((64, 64)), torch.from_numpy(d["B2"]).cuda())
    td = torch.from_numpy(lr_img.astype(np.float32).reshape(-1)).cuda()
    inr.SirenFitter(fitted, lr=1e-4).step(xd, td, 2500)
    ref = P.PortSiren(256, 512, 3, 1).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in fitted.state_dict().items()}, strict=False)
    net = inr.Siren(256, 512, 3, 1)
    net.load_state_dict({k: v.detach().cpu() for k, v in fitted.state_dict().items()}, strict=False)
    net.cuda()
    x, t, w = xd.cpu(), td.cpu(), None
else:
    head_scale = float(sys.argv[1]) if len(sys.argv) > 1 else 100.0
    net, ref = T.make_pair(256, 512, 3, seed=5)
    with torch.no_grad():
        net.final_linear.weight.mul_(head_scale)
        ref.final_linear.weight.mul_(head_scale)
    net.cuda()
    g = torch.Generator().manual_seed(23)
    n = 3000
    x = torch.rand(n, 256, generator=g) * 2 - 1
    t = torch.rand(n, generator=g)
    w = torch.rand(n, generator=g)
want_loss, want_g, _ = T.oracle_loss_grads(ref, x, t, w)
names = [f"{k}_{l}" for l in range(5) for k in ("W", "b")]
for label, keys in (("hl32 z-stash", {}), ("hl32 act+cos stash", {16: 0}), ("hl32 tile kernels", {10: 0}), ("h3 (round 1)", {7: 0}),
                    ("exact fp32", {3: 0})):
    for k, v in keys.items():
        ops.lib().inr_debug_set(k, v)
    loss, got, _ = T.fused_loss_grads(net, x.cuda(), t.cuda(), None if w is None else w.cuda())
    ops.lib().inr_debug_reset()
    print(f"{label:22s} loss err {abs(loss - want_loss) / want_loss:.1e}  " +
          " ".join(f"{nm}:{O.rel_l2(a, b):.1e}" for nm, a, b in zip(names, got, want_g)))
