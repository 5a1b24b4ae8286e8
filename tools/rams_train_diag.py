"""Per-layer gradient error of the RAMS training step against autograd on the float64 restatement."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import rams
from oracle import inr_oracle as O
from oracle import rams_port as R
N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
params = R.init_rams_params(seed=5, perturb_g=True, N=N)
model = rams.RAMS(3, 32, 3, 9, 8, N, params=params)
rng = np.random.default_rng(3)
B, side = 2, 20
x = (rng.random((B, side, side, 9)) * 20000 + 2000).astype(np.float32)
hr = (rng.random((B, 3 * side, 3 * side)) * 20000 + 2000).astype(np.float32)
mask = (rng.random((B, 3 * side, 3 * side)) > 0.15).astype(np.float32)
want_loss, want = R.train_grads(params, x, hr, mask, N=N)
tr = rams.RamsTrainer(model)
loss = tr.loss_and_grads(x, hr, mask)
got = tr.named_gradients()
print("loss", loss.cpu().numpy(), want_loss)
for name, _, _, _ in model.specs:
    print("%-22s" % name, " ".join("%s %.2e (|g| %.2e)" % (k, O.rel_l2(got[f"{name}/{k}"], want[f"{name}/{k}"]), np.linalg.norm(want[f"{name}/{k}"])) for k in "vgb"))
