"""RAMS forward under the profiler: `rocprofv3 --kernel-trace --stats -- python3 tools/rams_prof.py [batch]` (default model
RAMS(3,32,3,9,8,12), (B,128,128,9) stacks; one warm call + three profiled-alike calls)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import rams
model = rams.RAMS(seed=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
if len(sys.argv) > 2:          # debug key 15: which LDS-staged convolution kernel (42 = the product default: two blocks of 4 waves x 2 tiles per CU; 8, 4, 16 = the others)
    from mri_super_resolution_amd._lib import lib
    lib().inr_debug_set(15, int(sys.argv[2]))
xt = torch.from_numpy((np.random.default_rng(0).random((B, 128, 128, 9)) * 60000).astype(np.float32)).cuda()
for _ in range(4):
    model(xt)
torch.cuda.synchronize()
