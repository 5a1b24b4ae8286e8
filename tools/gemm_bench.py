import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mri_super_resolution_amd as inr
from mri_super_resolution_amd import ops, _lib
N=524288; H=512
lib=_lib.lib()
x=torch.randn(N,H,device='cuda'); W=torch.randn(H,H,device='cuda')*0.05; b=torch.randn(H,device='cuda')
dz=torch.randn(N,H,device='cuda'); dact=torch.randn(N,H,device='cuda')
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps
fl=2*N*H*H/1e9
for stag in [int(a) for a in sys.argv[1:]] or [0]:
    lib.inr_debug_set.argtypes=[ctypes.c_int,ctypes.c_int]; lib.inr_debug_set(0, stag)
    t1=timeit(lambda: ops.sine_layer_forward(x,W,b,30.0,True))
    t2=timeit(lambda: ops.sine_layer_backward_input(dz,W,dact))
    t3=timeit(lambda: ops.linear_param_grad(dz,x,False))
    print(f"generic={stag}: fwd {t1:.3f} ms {fl/t1:.1f} TF | dX {t2:.3f} ms {fl/t2:.1f} TF | dW {t3:.3f} ms {fl/t3:.1f} TF", flush=True)
