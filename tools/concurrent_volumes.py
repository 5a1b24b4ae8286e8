"""GPU box: `drivers.run_volumes(concurrent=k)` on a batch of config-1-sized fits (128 x 128 slices -> 4,096 training rows each) and of
small 3-D volumes: wall seconds and aggregate coordinate-steps/s for k = 1, 2, 3, 4."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mri_super_resolution_amd import drivers
rng = np.random.default_rng(0)
for name, shape, n in (("2-D slices 128 x 128", (128, 128), 12), ("volumes 128 x 128 x 4", (128, 128, 4), 8)):
    vols = [rng.random(shape).astype(np.float32) + 0.05 for _ in range(n)]
    steps = 500
    drivers.run_volumes(vols[:2], steps=20)                # warm
    base = None
    for k in (1, 2, 3, 4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        recs = drivers.run_volumes(vols, steps=steps, concurrent=k, evaluate=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        rows = sum(r["n_coords"] for r in recs) * steps
        base = base or dt
        print(f"{name}: {n} fits x {steps} steps, concurrent={k}: {dt:.2f} s, {rows / dt / 1e6:.1f} M coordinate-steps/s ({base / dt:.2f} x)", flush=True)
