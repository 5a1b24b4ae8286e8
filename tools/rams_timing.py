import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mri_super_resolution_amd import rams, ops
model = rams.RAMS(seed=0)
if len(sys.argv) > 1:      # arithmetic of the 32 -> 32 convolutions: 1 = split-fp16 (default), 0 = f32-input MFMA
    from mri_super_resolution_amd._lib import lib
    lib().inr_debug_set(14, int(sys.argv[1]))
    print('debug key 14 =', sys.argv[1])
    if len(sys.argv) > 2:
        lib().inr_debug_set(15, int(sys.argv[2]))
        print('debug key 15 =', sys.argv[2])
for B in (1, 5, 25):
    x = (np.random.default_rng(0).random((B, 128, 128, 9)) * 60000).astype(np.float32)
    xt = torch.from_numpy(x).cuda()
    model(xt); torch.cuda.synchronize()
    t0 = time.perf_counter(); reps = 3
    for _ in range(reps): model(xt)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"RAMS forward B={B}: {dt*1e3:.2f} ms  -> {265.0*B/dt/1e3:.1f} TFLOP/s (265 GFLOP per 128x128x9 stack), {B/dt:.1f} stacks/s")
