#!/usr/bin/env python3
"""Tier-T4 yardstick: the REAL reference (SRDWI.Siren / input_mapping / get_mgrid imported from /root/reference) run through
the full 2,500-step config-1 fit (pat07 slice 11, LR 64x64 -> HR 128x128), PSNR of the re-sampled 128x128 slice (clamped
at 0 as superresDWI.py:161-162 does) against the HR slice.  Seeds 0..11 were run on 8 CPU threads in round 2; round 3 adds
seeds 12..59 (2 threads each, several processes side by side -- the reference's result depends on the thread count only
through fp32 summation order) and, for those, the PSNR at a few steps before 2,500 as well
(evaluated under no_grad: the trajectory is untouched), so that a fit caught on an Adam spike at the last step can be told
from a biased one.

Build container only (2-6 minutes per seed); writes tests/golden/cfg1_ref_psnr.npz (numbers only).
    python oracle/gen_golden_t4.py first_seed last_seed [threads] [partial_out.npz]
    python oracle/gen_golden_t4.py --merge partial1.npz partial2.npz ...
"""
import os
import sys
import time
import warnings

import numpy as np
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "implicit-neural-representations"))
import SRDWI  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "cfg1_ref_psnr.npz")


TRACE_STEPS = (2300, 2350, 2400, 2450, 2480, 2490, 2495)


def _psnr(net, xt, hr):
    with torch.no_grad():
        sr = torch.clamp(net.forward(xt), min=0).view(hr.shape).numpy()
    mse = float(np.mean((sr.astype(np.float64) - hr.astype(np.float64)) ** 2))
    return 10.0 * np.log10(1.0 / mse)


def one_seed(hr, lr, seed, steps=2500, trace=False):
    np.random.seed(seed)
    torch.manual_seed(seed)
    B = torch.from_numpy(np.random.normal(size=(128, 2)) * 0.5).float()           # superresDWI.py:105-106
    net = SRDWI.Siren(in_features=256, out_features=1, hidden_features=512, hidden_layers=3)
    opt = torch.optim.Adam(lr=1e-4, params=list(net.parameters()))
    ds = SRDWI.ImageFitting_set([lr.astype(np.float64)])
    x = SRDWI.input_mapping(ds.coords[0], B)
    y = ds.pixels[0]
    xt = SRDWI.input_mapping(SRDWI.get_mgrid(hr.shape), B)
    tr = []
    for it in range(steps):                                                        # superresDWI.py:133-138
        if trace and it in TRACE_STEPS:
            tr.append(_psnr(net, xt, hr))
        out = net.forward(x)
        loss = ((out - y) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    return _psnr(net, xt, hr), float(loss), tr


def _load(path):
    if os.path.exists(path):
        d = dict(np.load(path))
        n = len(d["seeds"])
        if "trace_db" not in d:
            d["trace_db"] = np.full((n, len(TRACE_STEPS)), np.nan)
        if "threads_per_seed" not in d:
            d["threads_per_seed"] = np.full(n, 8, np.int64)
        return d
    return {"seeds": np.zeros(0, np.int64), "psnr_db": np.zeros(0), "final_loss": np.zeros(0),
            "trace_db": np.zeros((0, len(TRACE_STEPS))), "threads_per_seed": np.zeros(0, np.int64)}


def _save(path, rows):
    rows = sorted(rows, key=lambda r: r[0])
    np.savez(path, seeds=np.asarray([r[0] for r in rows], np.int64), psnr_db=np.asarray([r[1] for r in rows]),
             final_loss=np.asarray([r[2] for r in rows]), trace_db=np.asarray([r[3] for r in rows], np.float64),
             threads_per_seed=np.asarray([r[4] for r in rows], np.int64), trace_steps=np.asarray(TRACE_STEPS, np.int64),
             threads=np.int64(8), steps=np.int64(2500))


def _rows(d):
    return [(int(s), float(p), float(l), np.asarray(t, np.float64), int(th)) for s, p, l, t, th in
            zip(d["seeds"], d["psnr_db"], d["final_loss"], d["trace_db"], d["threads_per_seed"])]


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--merge":
        rows = {r[0]: r for r in _rows(_load(OUT))}
        for p in sys.argv[2:]:
            for r in _rows(_load(p)):
                rows.setdefault(r[0], r)
        _save(OUT, list(rows.values()))
        ps = np.asarray([r[1] for r in rows.values()])
        print("merged: %d seeds, mean %.3f sigma %.3f" % (len(ps), ps.mean(), ps.std(ddof=1)))
        return
    first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) >= 3 else (0, 11)
    threads = int(sys.argv[3]) if len(sys.argv) >= 4 else 8
    out = sys.argv[4] if len(sys.argv) >= 5 else OUT
    torch.set_num_threads(threads)
    g = np.load(os.path.join(HERE, "..", "tests", "golden", "pat07_slice11.npz"))
    hr, lr = g["hr"], g["lr"]
    rows = _rows(_load(out))
    done = {r[0] for r in rows} | ({r[0] for r in _rows(_load(OUT))} if out != OUT else set())
    for s in range(first, last + 1):
        if s in done:
            continue
        t0 = time.time()
        p, l, tr = one_seed(hr, lr, s, trace=True)
        rows.append((s, p, l, np.asarray(tr), threads))
        print(f"seed {s}: {p:.3f} dB, final loss {l:.3e}, {time.time() - t0:.0f} s", flush=True)
        _save(out, rows)
    ps = [r[1] for r in rows]
    print("mean %.3f  sigma %.3f over %d seeds" % (np.mean(ps), np.std(ps, ddof=1), len(ps)))


if __name__ == "__main__":
    main()
