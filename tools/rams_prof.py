import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import rams
model = rams.RAMS(seed=0, N=2)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
xt = torch.from_numpy((np.random.default_rng(0).random((B, 128, 128, 9)) * 60000).astype(np.float32)).cuda()
model(xt); model(xt); torch.cuda.synchronize()
