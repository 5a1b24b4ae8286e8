import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import ops
rng = np.random.default_rng(0)
for scale in (1.0, 8.0, 64.0, 1000.0, 6.0e4, 1.0e6, 3.0e6):
    x = ((rng.random(2000000) * 2 - 1) * scale).astype(np.float32)
    s, c = ops.sincos_probe(torch.from_numpy(x).cuda())
    xs = x.astype(np.float64)
    es = np.abs(s.cpu().numpy() - np.sin(xs)); ec = np.abs(c.cpu().numpy() - np.cos(xs))
    print(scale, "max", es.max(), ec.max(), "rms", np.sqrt((es**2).mean()), np.sqrt((ec**2).mean()))
