"""GPU box: A/B of whole libraries (INR_LIB) in SEPARATE processes, interleaved rounds: fused step at 128^3 (+ per-class GEMM times), the
exact-fp32 leg, 4,096-row step, dense re-sampling.  python tools/ab_libs.py product nopk ... (names = libinrhip_<name>.so; product = default)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import numpy as np, torch
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import drivers, ops
from mri_super_resolution_amd._lib import lib
out = {}
def step_ms(n, k, fp32=False):
    lib().inr_debug_set(3, 0 if fp32 else 1)
    g = torch.Generator(device="cuda").manual_seed(n)
    x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous(); t = torch.rand(n, device="cuda", generator=g)
    torch.manual_seed(0)
    f = inr.SirenFitter(inr.Siren(256, 512, 3, 1).cuda(), lr=1e-4)
    f.step(x, t, 3); torch.cuda.synchronize()
    ops.prof_reset(); ops.prof_enable(True)
    t0 = time.perf_counter(); f.step(x, t, k); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k * 1e3
    ops.prof_enable(False)
    cls = [ops.prof_read(c) for c in range(4)]
    lib().inr_debug_set(3, 1)
    return dt, [round(ms / max(nl, 1), 4) for nl, ms in cls[:3]] + [round(cls[3][1] / k, 4)]
def signature(n):      # a short deterministic fit: the bits two builds must share when a change claims "the same bits"
    import hashlib
    g = torch.Generator(device="cuda").manual_seed(7)
    x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous(); t = torch.rand(n, device="cuda", generator=g)
    torch.manual_seed(0)
    f = inr.SirenFitter(inr.Siren(256, 512, 3, 1).cuda(), lr=1e-4)
    losses = f.step(x, t, 12); torch.cuda.synchronize()
    return hashlib.sha256(f.flat.detach().cpu().numpy().tobytes() + losses.detach().cpu().numpy().tobytes()).hexdigest()[:12]
out["sig"] = [signature(4096), signature(150016)]
out["step128"], out["classes(fwd,dx,dw,other/step)"] = step_ms(524288, 30)
out["fp32_128"], _ = step_ms(524288, 8, True)
out["step4096"], _ = step_ms(4096, 300)
out["step69632"], _ = step_ms(69632, 100)
torch.manual_seed(0)
net = inr.Siren(256, 512, 3, 1).cuda(); B = torch.from_numpy(drivers.fourier_matrix(3, seed=0)).cuda()
inr.reconstruct(net, (256, 256, 128), B); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): inr.reconstruct(net, (256, 256, 128), B)
torch.cuda.synchronize(); out["recon_Mvox_s"] = 256 * 256 * 128 * 3 / (time.perf_counter() - t0) / 1e6
print(json.dumps(out))
''' % root
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ)
        env.pop("INR_LIB", None)
        if name != "product":
            env["INR_LIB"] = os.path.join(root, "mri-super-resolution_amd", f"libinrhip_{name}.so")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print(f"round {rnd} {name}:", line[-1] if line else r.stderr[-400:], flush=True)
