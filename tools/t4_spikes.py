"""Config-1 full-length fits, seeds 0-11: PSNR at step 2,500 and at the neighbouring steps, loss spikes over the last 500
steps, for the split-fp16 path and the exact-fp32 MFMA path (debug key 3).  A spike at the evaluation step is Adam at a
1e-7 loss level, not arithmetic: this prints the evidence."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import _lib, drivers, metrics
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pat07_slice11.npz"))
hr, lr = z["hr"], z["lr"]
hr_d = torch.from_numpy(np.ascontiguousarray(hr)).cuda().contiguous()
seeds = [int(a) for a in sys.argv[1:]] or list(range(12))
for mode, name in ((1, "split-fp16"), (0, "fp32 MFMA")):
    _lib.lib().inr_debug_set(3, mode)
    for s in seeds:
        torch.manual_seed(s)
        B = torch.from_numpy(drivers.fourier_matrix(2, seed=s)).cuda()
        net = inr.Siren(256, 512, 3, 1).cuda()
        ds = inr.ImageFitting_set([lr])
        x = inr.input_mapping(ds.coords[0], B)
        f = inr.SirenFitter(net, lr=1e-4)
        l = f.step(x, ds.pixels[0], 2480).cpu().numpy()
        ps, ls = [], [l]
        for k in range(8):                      # PSNR at steps 2480, 2485, ..., 2515
            ps.append(float(metrics.psnr(hr_d, inr.reconstruct(net, (128, 128), B).contiguous(), 1.0)))
            ls.append(f.step(x, ds.pixels[0], 5).cpu().numpy())
        l = np.concatenate(ls)
        last = l[2000:2500]
        print(f"{name:10s} seed {s:2d}: PSNR@2480..2515 {np.round(ps, 2)}  loss@2500 {l[2499]:.2e}  median(last 500) "
              f"{np.median(last):.2e}  max(last 500) {last.max():.2e}  spikes(>10x median) {(last > 10 * np.median(last)).sum()}", flush=True)
_lib.lib().inr_debug_set(3, 1)
