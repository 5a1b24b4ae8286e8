"""GPU box: RAMS forward (batch 25 and 1), debug key 24 on / off interleaved: 1 = long skip in the trunk-closing convolution's epilogue and
the temporal stages' convolutions writing the next block's padded input themselves; 0 = separate add / reflect-pad passes (same bits)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import rams
from mri_super_resolution_amd._lib import lib
model = rams.RAMS(seed=0)
for B, reps in ((25, 6), (1, 40)):
    xt = torch.from_numpy((np.random.default_rng(0).random((B, 128, 128, 9)) * 60000).astype(np.float32)).cuda()
    for rnd in range(3):
        for key in (2, 1, 0):
            lib().inr_debug_set(24, key)
            model(xt); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): model(xt)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
            print(f"batch {B} round {rnd} key24={key}: {dt * 1e3:.3f} ms, {265.0 * B / dt / 1e3:.1f} TFLOP/s", flush=True)
lib().inr_debug_set(24, 2)
