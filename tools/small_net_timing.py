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
