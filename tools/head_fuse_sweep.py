"""GPU box: fused-step time with the head step as its own kernel (key 30 = 0) and fused into the last sine layer's epilogue (key 30 = 1,
key 31 = 1: any row count), by row count -- where does the fused form start to pay?   python tools/head_fuse_sweep.py"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import mri_super_resolution_amd as inr
from mri_super_resolution_amd._lib import lib
ROWS = (32768, 65536, 69632, 98304, 114688, 131072, 139264, 196608, 262144, 393216, 524288, 1048576)
MODES = (("head step kernel", ((30, 0),)), ("head in the last layer's epilogue", ((30, 1), (31, 1))))
for n in ROWS:
    g = torch.Generator(device="cuda").manual_seed(n)
    x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous()
    t = torch.rand(n, device="cuda", generator=g)
    steps = max(12, min(200, int(1.0e7 / n)))
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
    row["fused / kernel"] = row[MODES[1][0]] / row[MODES[0][0]]
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()}), flush=True)
