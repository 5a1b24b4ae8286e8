import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
torch.manual_seed(0)
net = inr.Siren(2, 64, 6, 1).cuda()
coords = inr.get_mgrid(60, 2); tgt = torch.rand(3600, 1, device='cuda')
f = inr.SirenFitter(net, lr=3e-4)
f.step(coords, tgt, 200); torch.cuda.synchronize()
