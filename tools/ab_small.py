"""GPU box: A/B of debug-key settings on the fused fit step at small / mid row counts, interleaved rounds (best of them):
python tools/ab_small.py "rows,rows,..." "cfg" "cfg" ...   with cfg like "21=4,22=192" ("-" = defaults)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import _lib
lib = _lib.lib()
rows = [int(r) for r in sys.argv[1].split(",")]
cfgs = sys.argv[2:] or ["-"]
for n in rows:
    g = torch.Generator(device="cuda").manual_seed(n)
    x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous()
    t = torch.rand(n, device="cuda", generator=g)
    steps = max(40, min(400, int(2.0e7 / n)))
    best = {c: float("inf") for c in cfgs}
    for rep in range(3):
        for c in cfgs:
            lib.inr_debug_reset()
            if c != "-":
                for kv in c.split(","):
                    k, v = kv.split("=")
                    assert lib.inr_debug_set(int(k), int(v)) == 0, kv
            torch.manual_seed(0)
            f = inr.SirenFitter(inr.Siren(256, 512, 3, 1).cuda(), lr=1e-4)
            f.step(x, t, 5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            f.step(x, t, steps)
            torch.cuda.synchronize()
            best[c] = min(best[c], (time.perf_counter() - t0) / steps * 1e3)
            del f
    lib.inr_debug_reset()
    print(f"rows {n}: " + "  ".join(f"[{c}] {best[c]:.4f} ms" for c in cfgs), flush=True)
