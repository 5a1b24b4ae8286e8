"""GPU box: do small fits overlap when run from several host threads on streams of their own?  k threads, each a SirenFitter at n rows on
its own torch stream, fused steps in blocks of 50; aggregate coordinate-steps/s against one thread.  (Measurement only.)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mri_super_resolution_amd as inr

def worker(n, steps, stream, out, idx, barrier):
    with torch.cuda.stream(stream):
        g = torch.Generator(device="cuda").manual_seed(idx)
        x = (torch.rand(n, 256, device="cuda", generator=g) * 2 - 1).contiguous(); t = torch.rand(n, device="cuda", generator=g)
        torch.manual_seed(idx)
        f = inr.SirenFitter(inr.Siren(256, 512, 3, 1).cuda(), lr=1e-4)
        f.step(x, t, 5); stream.synchronize()
        barrier.wait()
        t0 = time.perf_counter()
        for _ in range(steps // 50):
            losses = f.step(x, t, 50)
        stream.synchronize()
        out[idx] = (time.perf_counter() - t0, float(losses[-1]))

for n in (4096, 16384):
    base = None
    for k in (1, 2, 4):
        steps = 1000
        out = [None] * k
        barrier = threading.Barrier(k)
        threads = [threading.Thread(target=worker, args=(n, steps, torch.cuda.Stream(), out, i, barrier)) for i in range(k)]
        for th in threads: th.start()
        for th in threads: th.join()
        wall = max(o[0] for o in out)
        rate = k * steps * n / wall
        base = base or rate
        print(f"rows {n}: {k} concurrent fits: {wall / steps * 1e3:.3f} ms per step each, aggregate {rate / 1e6:.1f} M coordinate-steps/s ({rate / base:.2f} x one fit); losses {[round(o[1], 6) for o in out]}", flush=True)
