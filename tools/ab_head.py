"""GPU box: head step of the fused fit at 128^3 -- rows per block (inr_debug_set(23, .)) x libraries (INR_LIB), separate processes,
interleaved.  Prints ms per step and the ms per step outside the GEMMs (head step + weight preparation + finalize).
python tools/ab_head.py product nopf"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import torch
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import ops
from mri_super_resolution_amd._lib import lib
out = {}
n = int(os.environ.get("AB_ROWS", "524288"))
g = torch.Generator(device="cuda").manual_seed(n)
x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous(); t = torch.rand(n, device="cuda", generator=g)
for rows in [int(v) for v in os.environ["AB_HEAD_ROWS"].split(",")]:
    lib().inr_debug_set(23, rows)
    torch.manual_seed(0)
    f = inr.SirenFitter(inr.Siren(256, 512, 3, 1).cuda(), lr=1e-4)
    f.step(x, t, 3); torch.cuda.synchronize()
    k = 30 if n > 100000 else 200
    ops.prof_reset(); ops.prof_enable(True)
    t0 = time.perf_counter(); losses = f.step(x, t, k); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k * 1e3
    ops.prof_enable(False)
    out[rows] = [round(dt, 4), round(ops.prof_read(3)[1] / k, 4), float(losses[-1]) if hasattr(losses, "__len__") else float(losses)]
    del f
print(json.dumps(out))
''' % root
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ)
        env.pop("INR_LIB", None)
        env.setdefault("AB_HEAD_ROWS", "0,64,128,172,344,512")
        if name != "product":
            env["INR_LIB"] = os.path.join(root, "mri-super-resolution_amd", f"libinrhip_{name}.so")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"round {rnd} {name}: rows -> [ms/step, other ms/step, last loss]", line[-1] if line else r.stderr[-600:], flush=True)
