"""GPU box: fused-step time at mid-sized row counts with the launches sent to the wide persistent kernels (default) or to the 64 x 128 tiles
of gemm_hp_nt_kernel (debug key 29 raised), and with the row-owning kernel (key 27 = 1, key 28 = 1).  python tools/narrow_sweep.py"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mri_super_resolution_amd as inr
from mri_super_resolution_amd._lib import lib
ROWS = (16384, 32768, 46421, 65536, 69632, 98304, 114688, 139264, 262144)
MODES = (("wide (default)", ()), ("narrow forced", ((29, 1 << 20),)), ("row-owning", ((27, 1), (28, 1), (18, 0))))
for n in ROWS:
    g = torch.Generator(device="cuda").manual_seed(n)
    x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous()
    t = torch.rand(n, device="cuda", generator=g)
    steps = max(20, min(300, int(1.5e7 / n)))
    row = {"rows": n}
    for rep in range(2):
        for name, keys in MODES:
            lib().inr_debug_reset()
            for k, v in keys:
                lib().inr_debug_set(k, v)
            torch.manual_seed(0)
            f = inr.SirenFitter(inr.Siren(256, 512, 3, 1).cuda(), lr=1e-4)
            f.step(x, t, 3); torch.cuda.synchronize()
            t0 = time.perf_counter(); f.step(x, t, steps); torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            row[name] = min(row.get(name, 1e9), ms)
    lib().inr_debug_reset()
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
