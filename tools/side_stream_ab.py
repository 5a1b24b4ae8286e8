#!/usr/bin/env python3
"""A/B of inr_debug_set(20, v): the parameter-gradient GEMMs of a step in ONE launch behind the input-gradient chain (1; the first
form of the switch put them on a second stream, hence the file name) or one launch per layer, in line (0); ms per fused step by
row count (`python tools/side_stream_ab.py [rows ...]`)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import mri_super_resolution_amd as inr  # noqa: E402
from mri_super_resolution_amd import ops  # noqa: E402

ROWS = tuple(int(a) for a in sys.argv[1:]) or (4096, 16384, 69632, 139264, 524288)
for n in ROWS:
    g = torch.Generator(device="cuda").manual_seed(n)
    x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous()
    t = torch.rand(n, device="cuda", generator=g)
    steps = max(20, min(400, int(2.0e7 / n)))
    out = []
    for key in (0, 1, 0, 1):
        with ops.debug_switch(20, key):
            torch.manual_seed(0)
            f = inr.SirenFitter(inr.Siren(256, 512, 3, 1).cuda(), lr=1e-4)
            f.step(x, t, 3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            losses = f.step(x, t, steps)
            torch.cuda.synchronize()
            out.append(((time.perf_counter() - t0) / steps * 1e3, float(losses[-1])))
    print(f"rows {n}: in line {out[0][0]:.4f} / {out[2][0]:.4f} ms, merged {out[1][0]:.4f} / {out[3][0]:.4f} ms; "
          f"final loss equal: {out[0][1] == out[1][1]}")
