#!/usr/bin/env python3
"""Tier-T4 yardstick: the REAL reference (SRDWI.Siren / input_mapping / get_mgrid imported from /root/reference) run through
the full 2,500-step config-1 fit (pat07 slice 11, LR 64x64 -> HR 128x128) for seeds 0..11 on 8 CPU threads, PSNR of the
re-sampled 128x128 slice (clamped at 0 as superresDWI.py:161-162 does) against the HR slice.

Build container only (about 2 minutes per seed); writes tests/golden/cfg1_ref_psnr.npz (numbers only).
    python oracle/gen_golden_t4.py [first_seed last_seed]
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


def one_seed(hr, lr, seed, steps=2500):
    np.random.seed(seed)
    torch.manual_seed(seed)
    B = torch.from_numpy(np.random.normal(size=(128, 2)) * 0.5).float()           # superresDWI.py:105-106
    net = SRDWI.Siren(in_features=256, out_features=1, hidden_features=512, hidden_layers=3)
    opt = torch.optim.Adam(lr=1e-4, params=list(net.parameters()))
    ds = SRDWI.ImageFitting_set([lr.astype(np.float64)])
    x = SRDWI.input_mapping(ds.coords[0], B)
    y = ds.pixels[0]
    for _ in range(steps):                                                         # superresDWI.py:133-138
        out = net.forward(x)
        loss = ((out - y) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    xt = SRDWI.input_mapping(SRDWI.get_mgrid(hr.shape), B)
    sr = torch.clamp(net.forward(xt), min=0).view(hr.shape).detach().numpy()
    mse = float(np.mean((sr.astype(np.float64) - hr.astype(np.float64)) ** 2))
    return 10.0 * np.log10(1.0 / mse), float(loss)


def main():
    torch.set_num_threads(8)
    first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) == 3 else (0, 11)
    g = np.load(os.path.join(HERE, "..", "tests", "golden", "pat07_slice11.npz"))
    hr, lr = g["hr"], g["lr"]
    have = dict(np.load(OUT)) if os.path.exists(OUT) else {"seeds": np.zeros(0, np.int64), "psnr_db": np.zeros(0), "final_loss": np.zeros(0)}
    seeds, psnr, fl = list(have["seeds"]), list(have["psnr_db"]), list(have["final_loss"])
    for s in range(first, last + 1):
        if s in seeds:
            continue
        t0 = time.time()
        p, l = one_seed(hr, lr, s)
        seeds.append(s), psnr.append(p), fl.append(l)
        print(f"seed {s}: {p:.3f} dB, final loss {l:.3e}, {time.time() - t0:.0f} s", flush=True)
        order = np.argsort(seeds)
        np.savez(OUT, seeds=np.asarray(seeds, np.int64)[order], psnr_db=np.asarray(psnr)[order],
                 final_loss=np.asarray(fl)[order], threads=np.int64(8), steps=np.int64(2500))
    print("mean %.3f  sigma %.3f over %d seeds" % (np.mean(psnr), np.std(psnr, ddof=1), len(psnr)))


if __name__ == "__main__":
    main()
