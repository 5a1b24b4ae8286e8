#!/usr/bin/env python3
"""Tier T4, both arithmetic paths, every reference seed.  The full 2,500-step config-1 fit (pat07 slice 11,
superresDWI.py:105-138,161-162) for the 60 seeds the REAL reference was run at (tests/golden/cfg1_ref_psnr.npz) on
  * split_fp16  -- the product path (HL32 operands, 3 fp16 MFMA products per fp32 product, a-priori dz scales; key 3 = 1),
  * exact_fp32  -- the f32-input MFMA kernels (key 3 = 0),
  * split_h3    -- round 1's split kernels (fp32 operands in HBM, split in the consumer, EXACT max|dz| scales; key 7 = 0): separates
                   "three fp16 products" from "scales taken from an a-priori bound",
tracking per seed: PSNR at the trace steps and at 2,500, the loss of every step, max loss over the last 500 steps, the number of
spike steps there (> 10 x the median of those 500), whether step 2,500 itself is a spike.  Paired statistics over seeds
(Wilcoxon signed-rank + sign counts) say whether the split path spikes more often / higher / ends on a spike more often.
    python tools/t4_paths.py [out.json] [max_seeds]
"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy import stats as sst  # noqa: E402

import mri_super_resolution_amd as inr  # noqa: E402
from mri_super_resolution_amd import drivers, metrics, ops  # noqa: E402

TRACE_STEPS = (2300, 2350, 2400, 2450, 2480, 2490, 2495, 2500)
ARMS = (("split_fp16", ((3, 1),)), ("exact_fp32", ((3, 0),)), ("split_h3", ((3, 1), (7, 0))))


def one_seed(hr_t, lr, seed):
    torch.manual_seed(seed)
    B = torch.from_numpy(drivers.fourier_matrix(2, seed=seed)).cuda()
    net = inr.Siren(256, 512, 3, 1).cuda()
    ds = inr.ImageFitting_set([lr])
    x = inr.input_mapping(ds.coords[0], B)
    fitter = inr.SirenFitter(net, lr=1e-4)
    tr, done, losses = [], 0, []
    for upto in TRACE_STEPS:
        losses.append(fitter.step(x, ds.pixels[0], upto - done))
        done = upto
        tr.append(float(metrics.psnr(hr_t, inr.reconstruct(net, tuple(hr_t.shape), B), 1.0)))
    loss = torch.cat(losses).cpu().numpy().astype(np.float64)
    tail = loss[-500:]
    med = float(np.median(tail))
    return {"trace_db": tr, "psnr_db": tr[-1], "final_loss": float(loss[-1]), "tail_median": med, "tail_max": float(tail.max()),
            "spikes": int((tail > 10 * med).sum()), "big_spikes": int((tail > 100 * med).sum()), "ends_on_spike": bool(loss[-1] > 10 * med),
            "loss_at": {str(k): float(loss[k - 1]) for k in (1, 10, 100, 500, 1000, 1500, 2000, 2500)}}


def two_sample(ours, ref):
    ours, ref = np.asarray(ours, np.float64), np.asarray(ref, np.float64)
    d = ours.mean() - ref.mean()
    se = float(np.sqrt(ours.var(ddof=1) / len(ours) + ref.var(ddof=1) / len(ref)))
    return {"mean": float(ours.mean()), "sigma": float(ours.std(ddof=1)), "ref_mean": float(ref.mean()), "ref_sigma": float(ref.std(ddof=1)),
            "delta": float(d), "se": se, "delta_over_se": float(d / se) if se else None}


def paired(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = a - b
    nz = d[d != 0]
    w = sst.wilcoxon(nz) if len(nz) >= 6 else None
    return {"mean_diff": float(d.mean()), "se": float(d.std(ddof=1) / np.sqrt(len(d))), "a_greater": int((d > 0).sum()), "b_greater": int((d < 0).sum()),
            "ties": int((d == 0).sum()), "wilcoxon_p": float(w.pvalue) if w is not None else None}


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05_t4_paths.json")
    g = np.load(os.path.join(ROOT, "tests", "golden", "pat07_slice11.npz"))
    ref = np.load(os.path.join(ROOT, "tests", "golden", "cfg1_ref_psnr.npz"))
    seeds = [int(s) for s in ref["seeds"]][:int(sys.argv[2]) if len(sys.argv) > 2 else None]
    hr, lr = np.ascontiguousarray(g["hr"], np.float32), np.ascontiguousarray(g["lr"], np.float32)
    hr_t = torch.from_numpy(hr).cuda()
    ref_final = {int(s): float(p) for s, p in zip(ref["seeds"], ref["psnr_db"])}
    out = {"seeds": seeds, "trace_steps": list(TRACE_STEPS), "arms": {}, "reference_psnr_db": [ref_final[s] for s in seeds]}
    for name, keys in ARMS:
        from mri_super_resolution_amd._lib import check, lib
        for k, v in keys:
            check(lib().inr_debug_set(k, v), "inr_debug_set")
        per = []
        try:
            for s in seeds:
                per.append(one_seed(hr_t, lr, s))
        finally:
            lib().inr_debug_reset()
        out["arms"][name] = per
        ps = [r["psnr_db"] for r in per]
        med = [float(np.median(r["trace_db"])) for r in per]
        print(f"{name}: PSNR@2500 {np.mean(ps):.3f} +- {np.std(ps, ddof=1):.3f}; median-over-trace {np.mean(med):.3f}; "
              f"spike steps / 500: {np.mean([r['spikes'] for r in per]):.1f}; >100x: {np.mean([r['big_spikes'] for r in per]):.2f}; "
              f"log10 max tail loss {np.mean(np.log10([r['tail_max'] for r in per])):.2f}; ends on a spike: "
              f"{sum(r['ends_on_spike'] for r in per)} of {len(per)}", flush=True)
    A = out["arms"]
    summ = {}
    for name in A:
        ps = [r["psnr_db"] for r in A[name]]
        summ[name] = {"psnr_vs_reference_plain": two_sample(ps, out["reference_psnr_db"]),
                      "psnr_vs_reference_trim5": two_sample(sorted(ps)[len(ps) // 20:len(ps) - len(ps) // 20] if len(ps) >= 20 else ps,
                                                            sorted(out["reference_psnr_db"])[len(ps) // 20:len(ps) - len(ps) // 20] if len(ps) >= 20 else out["reference_psnr_db"]),
                      "ends_on_spike": int(sum(r["ends_on_spike"] for r in A[name])),
                      "seeds_below_31_db": [s for s, r in zip(seeds, A[name]) if r["psnr_db"] < 31.0]}
    ref_low = int(sum(1 for s in seeds if ref_final[s] < 31.0))
    summ["reference"] = {"seeds_below_31_db": [s for s in seeds if ref_final[s] < 31.0], "count": ref_low}
    cmp = {}
    for a, b in (("split_fp16", "exact_fp32"), ("split_h3", "exact_fp32"), ("split_fp16", "split_h3")):
        cmp[f"{a}_minus_{b}"] = {
            "psnr_db": paired([r["psnr_db"] for r in A[a]], [r["psnr_db"] for r in A[b]]),
            "psnr_median_over_trace_db": paired([np.median(r["trace_db"]) for r in A[a]], [np.median(r["trace_db"]) for r in A[b]]),
            "log10_tail_max": paired(np.log10([r["tail_max"] for r in A[a]]), np.log10([r["tail_max"] for r in A[b]])),
            "log10_tail_median": paired(np.log10([r["tail_median"] for r in A[a]]), np.log10([r["tail_median"] for r in A[b]])),
            "spike_steps": paired([r["spikes"] for r in A[a]], [r["spikes"] for r in A[b]]),
            "big_spike_steps": paired([r["big_spikes"] for r in A[a]], [r["big_spikes"] for r in A[b]])}
        # end-on-spike rate: exact binomial (Fisher) on the 2 x 2 table
        ea, eb = sum(r["ends_on_spike"] for r in A[a]), sum(r["ends_on_spike"] for r in A[b])
        cmp[f"{a}_minus_{b}"]["ends_on_spike"] = {"a": int(ea), "b": int(eb), "n": len(seeds),
                                                  "fisher_p": float(sst.fisher_exact([[ea, len(seeds) - ea], [eb, len(seeds) - eb]])[1])}
    out["summary"], out["paired"] = summ, cmp
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps({"summary": summ, "paired": cmp}, indent=1))
    print("wrote", out_path)


if __name__ == "__main__":
    main()
