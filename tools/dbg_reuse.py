import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import mri_super_resolution_amd as inr
for n in (300, 360, 600):
    for hid in (64, 128):
        torch.manual_seed(0)
        net = inr.Siren(32, hid, 1, 1).cuda()
        x = (torch.rand(n, 32) * 2 - 1).cuda(); t = torch.rand(n, 1).cuda()
        f = inr.ShardedSirenFitter(net, global_rows=n, lr=1e-4)
        try:
            l = f.step(x, t, 3)
            print(n, hid, "ok", l.cpu().numpy())
        except Exception as e:
            print(n, hid, "ERR", str(e)[:200])
