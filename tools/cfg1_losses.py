import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import _lib
from oracle import torch_port as P, inr_oracle as O
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pat07_slice11.npz")); hr, lr = z["hr"], z["lr"]
B = torch.from_numpy(P.fourier_matrix(2)).cuda()
for mode in (1, 0):
    _lib.lib().inr_debug_set(1, mode)
    torch.manual_seed(0)
    net = inr.Siren(256, 512, 3, 1).cuda()
    ds = inr.ImageFitting_set([lr]); x = inr.input_mapping(ds.coords[0], B)
    f, losses = inr.fit_siren(net, x, ds.pixels[0], 2500, lr=1e-4)
    l = losses.cpu().numpy()
    rec = inr.reconstruct(net, (128, 128), B).cpu().numpy()
    print("mfma16=%d psnr %.3f  loss[::250] %s  last5 %s  max(last 200) %.2e median(last 200) %.2e" % (
        mode, O.psnr(hr, rec), np.array2string(l[::250], precision=2), np.array2string(l[-5:], precision=2), l[-200:].max(), np.median(l[-200:])))
_lib.lib().inr_debug_set(1, 1)
