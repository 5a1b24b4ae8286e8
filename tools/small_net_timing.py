import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import drivers
rng = np.random.default_rng(0)
acqs = [rng.random((60, 60)).astype(np.float32) for _ in range(8)]
wts = [np.ones((60, 60), np.float32) for _ in range(8)]
drivers.fit_slice_ensemble(acqs, wts, total_steps=5, seg=2)
r = drivers.fit_slice_ensemble(acqs, wts, total_steps=300, seg=150)
print("master.py regime: %d optimizer steps in %.3f s -> %.1f us/step, %.2f M coord-steps/s" % (
    r["optimizer_steps"], r["seconds"], r["seconds"] / r["optimizer_steps"] * 1e6, r["train_voxels_per_s"] / 1e6))
# device-side cost: many steps on ONE acquisition through a single C call
import mri_super_resolution_amd as inr
torch.manual_seed(0)
net = inr.Siren(2, 64, 6, 1).cuda()
coords = inr.get_mgrid(60, 2); tgt = torch.rand(3600, 1, device='cuda')
f = inr.SirenFitter(net, lr=3e-4)
f.step(coords, tgt, 50); torch.cuda.synchronize()
t0 = time.perf_counter(); f.step(coords, tgt, 2000); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("single C call, 2000 steps: %.1f us/step (%.2f M coord-steps/s)" % (dt / 2000 * 1e6, 3600 * 2000 / dt / 1e6))
