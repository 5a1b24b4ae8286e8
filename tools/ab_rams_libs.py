"""GPU box: RAMS forward (batch 1 / 25) and training step, whole libraries (INR_LIB) in separate processes, interleaved rounds.
python tools/ab_rams_libs.py product nopk"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import numpy as np, torch
from mri_super_resolution_amd import rams
out = {}
model = rams.RAMS(seed=0)
for B in (1, 25):
    xt = torch.from_numpy((np.random.default_rng(0).random((B, 128, 128, 9)) * 60000).astype(np.float32)).cuda()
    model(xt); torch.cuda.synchronize()
    reps = 20 if B == 1 else 4
    t0 = time.perf_counter()
    for _ in range(reps): model(xt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    out[f"fwd_b{B}_ms"] = round(dt * 1e3, 3); out[f"fwd_b{B}_tflops"] = round(265.0 * B / dt / 1e3, 1)
B, P = 32, 32
rng = np.random.default_rng(0)
tr = rams.RamsTrainer(rams.RAMS(seed=0))
lr = (rng.random((B, P, P, 9)) * 20000).astype(np.float32); hr = (rng.random((B, 3 * P, 3 * P, 1)) * 20000).astype(np.float32)
mask = np.ones((B, 3 * P, 3 * P, 1), np.float32)
tr.train_step(lr, hr, mask); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): tr.train_step(lr, hr, mask)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
out["train_ms"] = round(dt * 1e3, 3); out["train_tflops"] = round(3 * 265e9 * (P * P) / (128 * 128) * B / dt / 1e12, 1)
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
        print(f"round {rnd} {name}:", line[-1] if line else r.stderr[-600:], flush=True)
