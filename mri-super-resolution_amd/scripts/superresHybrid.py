#!/usr/bin/env python3
"""``superresHybrid.py`` of the reference (implicit-neural-representations/superresHybrid.py) on the MI355X path.

``master.mat`` (``hybrid_raw``: [b][TE] cell -- b = 0: [X, Y, Z], b > 0: [X, Y, Z, acquisitions] --, ``b``, ``TE``) in; per echo
time one 4-D (x, y, z, b) INR fit of the ROI on every second in-plane voxel, re-sampled at twice the ROI size
(superresHybrid.py:51-125); the 16 re-scaled images normalised by the (b = 0, TE = 0) one and one slice pushed through the
three-compartment fit ``PIA.hybrid_fit`` (:127-140); out: ``recon_hybrid`` [2 sx, 2 sy, Z, 4, 4], the compartment maps ``D``,
``T2``, ``v`` [2 sx, 2 sy, 3], the ADC map of the slice (:173) and the predicted cancer map ``(v_ep > 0.4) & (v_lu <= 0.2)`` with
objects under 12 pixels removed (:163-178) as ``hybrid.mat`` + ``recon_hybrid.npy``, the (header-only, as in the reference)
``ssim_scores.csv`` and a ``metrics.json``.  The reference hard-codes patient, paths and hyper-parameters (:26-37,65-75); the
same names are flags here with the same defaults.  Figures (:142-187) are outside the build's scope.

Under ``torchrun`` (one process per GPU) the four echo-time fits are spread over the ranks (``drivers.fit_hybrid(distributed=
True)``: 4 ranks one fit each, 8 ranks pairs that row-shard their fit; one all-reduce of the fitted slice); rank 0 writes.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))
import mri_super_resolution_amd as inr  # noqa: E402
from mri_super_resolution_amd import drivers, matio  # noqa: E402

SSIM_HEADER_HYBRID = "Pt_id, b-value, te-value, slice, SSIM-spline, SSIM-SR\n"        # superresHybrid.py:27


def build_parser():
    p = argparse.ArgumentParser(description="Hybrid multi-dimensional MRI: per-TE INR super-resolution + three-compartment fit")
    p.add_argument("--data", required=True, help="master.mat with hybrid_raw ([b][TE] cell), b and TE")
    p.add_argument("--pt_id", default=None)
    p.add_argument("--output_address", default="SR_Hybrid_results")
    p.add_argument("--number_of_epochs", type=int, default=2500)
    p.add_argument("--hidden_dim", type=int, default=512)
    p.add_argument("--num_layers", type=int, default=3)
    p.add_argument("--mapping_size", type=int, default=128)
    p.add_argument("--scale", type=float, default=0.5, help="sigma of the Gaussian Fourier features")
    p.add_argument("--roi_start_x", type=int, default=35)
    p.add_argument("--roi_end_x", type=int, default=95)
    p.add_argument("--roi_start_y", type=int, default=35)
    p.add_argument("--roi_end_y", type=int, default=95)
    p.add_argument("--slice", type=int, default=9, dest="_slice", help="z-slice of the three-compartment fit (:127)")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--fp16_targets", action="store_true", help="hold the normalised volume in half precision (BASELINE config 5)")
    return p


def load_hybrid(path):
    """-> (hybrid_raw [X, Y, Z, 4, 4] float32: acquisitions of b > 0 averaged (superresHybrid.py:51-54), bvalues, TE values)."""
    data = matio.loadmat(path)
    cell = data["hybrid_raw"]
    nb, nte = cell.shape
    vols = [[np.asarray(cell[b][te], np.float64) for te in range(nte)] for b in range(nb)]
    vols = [[v.mean(axis=-1) if (b and v.ndim == 4) else v for v in row] for b, row in enumerate(vols)]
    raw = np.stack([np.stack(row, axis=-1) for row in vols], axis=-2)                    # [X, Y, Z, b, te]
    bvals = np.asarray(data["b"], np.float64).reshape(-1) if "b" in data else np.array([0.0, 150.0, 1000.0, 1500.0])
    te = np.asarray(data["TE"], np.float64).reshape(-1) if "TE" in data else np.arange(nte, dtype=np.float64)
    return raw.astype(np.float32), bvals, te


def remove_small_objects(mask, min_size=12):
    """skimage.morphology.remove_small_objects(mask, min_size, connectivity=1) (superresHybrid.py:177): 4-connected components
    smaller than ``min_size`` pixels are cleared."""
    from scipy import ndimage as ndi
    lab, n = ndi.label(mask, structure=ndi.generate_binary_structure(mask.ndim, 1))
    if n == 0:
        return mask.copy()
    sizes = np.bincount(lab.ravel())
    keep = sizes >= min_size
    keep[0] = False
    return keep[lab]


def run(args):
    spread = int(os.environ.get("WORLD_SIZE", "1")) > 1
    rank = 0
    if spread:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(os.environ.get("INR_BACKEND", "nccl"))
        rank = dist.get_rank()
    raw, bvalues, te_values = load_hybrid(args.data)
    pt_id = args.pt_id or "".join(c for c in os.path.basename(os.path.dirname(os.path.abspath(args.data))) if c.isdigit()) or "0"
    roi = (args.roi_start_x, args.roi_end_x, args.roi_start_y, args.roi_end_y)
    if not (0 <= roi[0] < roi[1] <= raw.shape[0] and 0 <= roi[2] < roi[3] <= raw.shape[1]):
        raise ValueError(f"ROI {roi} does not fit the {raw.shape[:2]} slices")
    if not 0 <= args._slice < raw.shape[2]:
        raise ValueError(f"--slice {args._slice} outside the {raw.shape[2]} slices")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = drivers.fit_hybrid(raw, roi=roi, slice_index=args._slice, steps=args.number_of_epochs, seed=args.seed,
                             distributed=spread, gather_recon=spread, hidden_features=args.hidden_dim,
                             hidden_layers=args.num_layers, mapping_size=args.mapping_size, ff_scale=args.scale,
                             target_dtype=np.float16 if args.fp16_targets else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank != 0:
        return None
    out_dir = os.path.join(args.output_address, f"pat{int(pt_id):03d}" if str(pt_id).isdigit() else f"pat{pt_id}")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "ssim_scores.csv"), "w") as fh:                      # superresHybrid.py:40-41 (header only)
        fh.write(SSIM_HEADER_HYBRID)
    recon = res["recon_hybrid"].cpu().numpy()
    bins = res["v"].shape[:2]
    v_ep, v_lu = res["v"][..., 0], res["v"][..., 2]
    cancer = remove_small_objects((v_ep > 0.4) & (v_lu <= 0.2), 12)                       # :163-178
    adc_map = inr.calculate_ADC(bvalues, np.squeeze(recon[:, :, args._slice, :, 0]))      # :173
    matio.savemat(os.path.join(out_dir, "hybrid.mat"), {
        "recon_hybrid": recon, "D": res["D"], "T2": res["T2"], "v": res["v"], "status": res["status"].astype(np.int32),
        "adc_map": adc_map, "cancer_map": cancer.astype(np.uint8), "b": bvalues, "TE": te_values,
        "slice": np.array([args._slice], np.int32)})
    np.save(os.path.join(out_dir, "recon_hybrid.npy"), recon)
    summary = {"pt_id": str(pt_id), "input": os.path.abspath(args.data), "roi": list(roi), "slice": int(args._slice),
               "recon_shape": list(recon.shape), "map_shape": list(bins), "steps": int(args.number_of_epochs),
               "seconds": dt, "t_fit_s": res["t_fit"], "t_recon_s": res["t_recon"], "t_hybrid_fit_s": res["t_hybrid_fit"],
               "final_losses": res["final_losses"], "cancer_pixels": int(cancer.sum()),
               "voxel_fits_converged": float((res["status"] > 0).mean())}
    with open(os.path.join(out_dir, "metrics.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))
    return summary


def main(argv=None):
    return run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
