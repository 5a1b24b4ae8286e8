"""Phase clocks of the LDS-staged RAMS convolution (GPU box; -DR3_STAMPS build):
    python tools/rams_ablate_build.py stamps 1 && INR_LIB=mri-super-resolution_amd/libinrhip_r3stamps1.so python tools/rams_stamps.py [batch]
Every wave sums its s_memtime cycles per phase of the patch loop; printed: share of the summed wave cycles and cycles per patch
and wave (61 patches per wave at batch 25)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import rams
from mri_super_resolution_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
model = rams.RAMS(seed=0)
xt = torch.from_numpy((np.random.default_rng(0).random((B, 128, 128, 9)) * 60000).astype(np.float32)).cuda()
names = ["wait: image free", "split + LDS write", "post + global loads", "wait: image ready", "tile decode", "taps",
         "epilogue"]
for key15 in (8, 42):
    lib().inr_debug_set(15, key15)
    model(xt); torch.cuda.synchronize()
    st = torch.zeros(16, dtype=torch.int64, device="cuda")
    lib().inr_debug_set_ptr(0, st.data_ptr())
    model(xt); torch.cuda.synchronize()
    lib().inr_debug_set_ptr(0, None)
    v = st.cpu().numpy().astype(np.float64)
    waves, tot = v[8], v[:7].sum()
    per_block = 8 if key15 == 8 else 4
    npatch = 32 * 19
    blocks = (256 if key15 == 8 else 512) // B
    iters = -(-npatch // blocks)
    print(f"key 15 = {key15}: {int(waves)} waves stamped, {tot / waves / iters:.0f} cycles per patch and wave")
    for n, c in zip(names, v[:7]):
        print(f"   {n:22s} {100 * c / tot:5.1f} %   {c / waves / iters:8.0f} cycles per patch")
