"""Time of config 1 (pat07 slice 11: N = 4,096 rows, Siren(256,512,3,1), 2,500 steps) and of a 32,768-row fit."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import ops
from oracle import torch_port as P
for shape in ((64, 64), (128, 128), (256, 128)):
    B = torch.from_numpy(P.fourier_matrix(2)).cuda()
    x = ops.grid_fourier_map(shape, B)
    t = torch.rand(x.shape[0], device="cuda")
    torch.manual_seed(0)
    net = inr.Siren(256, 512, 3, 1).cuda()
    fit = inr.SirenFitter(net)
    fit.step(x, t, 50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fit.step(x, t, 500)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 500
    print(f"N = {x.shape[0]:6d}: {dt*1e6:7.1f} us/step  {x.shape[0]/dt/1e6:6.2f} M coordinate-steps/s", flush=True)
