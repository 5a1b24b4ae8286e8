#!/usr/bin/env python3
"""``superresDWI.py`` of the reference (implicit-neural-representations/superresDWI.py) on the MI355X path.

``.mat`` in -> ``recon`` (x2 the HR in-plane grid = "x4" w.r.t. the LR training grid) and ``SR_recon`` (HR grid) out as
``.mat`` + ``.npy``, plus ``ssim_scores.csv`` with the reference's header ``Pt_id, b-value, slice, SSIM-spline, SSIM-SR``
(superresDWI.py:27,186-187) and a PSNR/SSIM summary (``metrics.json``).

The reference hard-codes its patient list, paths and hyper-parameters (superresDWI.py:26-34,84-91); here the same NAMES are
flags with the same defaults: ``--number_of_epochs 2500 --pertubation_epochs 10 --hidden_dim 512 --num_layers 3 --PN_dim 128
--roi_start 40 --roi_end 90 --mapping_size 128 --scale 0.5``.  Inputs:
  * a ``master.mat`` with ``hybrid_raw`` ([b][TE] cell of [X, Y, Z, acquisitions]) and ``b`` -- the reference's format: the
    acquisition products, the mean image, the INR fit, the PerturbNet phase (superresDWI.py:44-156);
  * any ``.mat`` holding one volume [X, Y, Z] or [X, Y, Z, b] (e.g. ``anon_data/patNN_mean_b0.mat``, key ``data_mean_b0``):
    the same fit and evaluation without a PerturbNet phase (there are no single acquisitions to perturb towards).
Plots (superresDWI.py:164-233) are outside the build's scope.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(_HERE)))
import mri_super_resolution_amd as inr  # noqa: E402
from mri_super_resolution_amd import baselines, drivers, matio, metrics, reports  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser(description="INR super-resolution of diffusion MRI volumes (superresDWI.py protocol)")
    p.add_argument("--data", nargs="+", required=True, help=".mat file(s): master.mat (hybrid_raw) or a plain volume")
    p.add_argument("--key", default=None, help="variable holding the volume (default: hybrid_raw, else the only array)")
    p.add_argument("--pt_id", nargs="*", default=None, help="patient ids for the CSV (default: digits of the file name)")
    p.add_argument("--output_address", default="SR_results", help="output directory (one sub-directory per patient)")
    p.add_argument("--number_of_epochs", type=int, default=2500)
    p.add_argument("--pertubation_epochs", type=int, default=10)
    p.add_argument("--hidden_dim", type=int, default=512)
    p.add_argument("--num_layers", type=int, default=3)
    p.add_argument("--PN_dim", type=int, default=128)
    p.add_argument("--roi_start", type=int, default=40)
    p.add_argument("--roi_end", type=int, default=90)
    p.add_argument("--mapping_size", type=int, default=128)
    p.add_argument("--scale", type=float, default=0.5, help="sigma of the Gaussian Fourier features")
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--seed", type=int, default=None, help="seeds numpy (Fourier matrix) and torch (weights); default: unseeded")
    return p


def _patient_id(path):
    m = re.findall(r"\d+", os.path.basename(path))
    return m[0] if m else os.path.splitext(os.path.basename(path))[0]


def load_input(path, key=None):
    """-> (mean_img [X, Y, Z, B] float64, acquisitions or None, bvalues, maxes or None)."""
    data = matio.loadmat(path)
    if key is None and "hybrid_raw" in data:
        key = "hybrid_raw"
    if key is None:
        keys = [k for k, v in data.items() if isinstance(v, np.ndarray) and v.dtype != object and v.ndim >= 3]
        if len(keys) != 1:
            raise KeyError(f"{path}: pass --key, candidates are {keys}")
        key = keys[0]
    arr = data[key]
    if arr.dtype == object:                                           # hybrid_raw: superresDWI.py:44-83
        raw = [[np.asarray(arr[b][te], np.float64) for te in range(arr.shape[1])] for b in range(arr.shape[0])]
        maxes = np.array([[raw[b][te].max() for te in range(len(raw[b]))] for b in range(len(raw))])
        norm = [[raw[b][te] / maxes[b, te] for te in range(len(raw[b]))] for b in range(len(raw))]
        acq = drivers.acquisition_products(norm)                       # [X, Y, Z, B, K]
        bvals = np.asarray(data["b"], np.float64).reshape(-1) if "b" in data else np.arange(acq.shape[3], dtype=np.float64)
        return acq.mean(axis=-1), acq, bvals, maxes
    vol = np.asarray(arr, np.float64)
    if vol.ndim == 3:
        vol = vol[..., None]
    vol = vol / vol.reshape(-1, vol.shape[-1]).max(axis=0)             # per-b normalisation (superresDWI.py:50-55)
    bvals = np.asarray(data["b"], np.float64).reshape(-1) if "b" in data else np.zeros(vol.shape[-1])
    return vol, None, bvals, None


def run_patient(path, pt_id, args):
    out_dir = os.path.join(args.output_address, f"pat{pt_id}")
    os.makedirs(out_dir, exist_ok=True)
    mean_img, acq, bvalues, maxes = load_input(path, args.key)
    r0, r1 = args.roi_start, args.roi_end
    if r1 > min(mean_img.shape[:2]) or r0 < 0 or r1 - r0 < 14:
        raise ValueError(f"ROI {r0}:{r1} does not fit the {mean_img.shape[:2]} slices (SSIM needs >= 7 x 7 LR pixels)")
    if args.seed is not None:
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    lr_img = mean_img[r0:r1:2, r0:r1:2]                                                   # superresDWI.py:94,97
    hr_img = mean_img[r0:r1, r0:r1]                                                       # :100,128
    mean_dataset = inr.ImageFitting_set([lr_img])
    dimension = len(mean_dataset.shape)
    B = torch.from_numpy(np.random.normal(size=(args.mapping_size, dimension)) * args.scale).float().cuda()   # :105-106
    INR = inr.Siren(in_features=2 * args.mapping_size, out_features=1, hidden_features=args.hidden_dim,
                    hidden_layers=args.num_layers).cuda()
    model_input = inr.input_mapping(mean_dataset.coords[0], B)
    target = mean_dataset.pixels[0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if acq is None:
        fitter, losses = inr.fit_siren(INR, model_input, target, args.number_of_epochs, lr=args.learning_rate)
    else:
        acq_lr = [acq[r0:r1:2, r0:r1:2, :, :, k] for k in range(acq.shape[-1])]
        losses = drivers.fit_with_perturbnet(INR, B, mean_dataset, acq_lr, args.number_of_epochs, args.pertubation_epochs,
                                             PN_dim=args.PN_dim, lr=args.learning_rate)
    torch.cuda.synchronize()
    t_fit = time.perf_counter() - t0
    hr_shape = tuple(hr_img.shape)
    test_shape = (hr_shape[0] * 2, hr_shape[1] * 2) + hr_shape[2:]                         # :125
    t0 = time.perf_counter()
    recon = inr.reconstruct(INR, test_shape, B)                                           # :161
    SR_recon = inr.reconstruct(INR, hr_shape, B)                                          # :162
    torch.cuda.synchronize()
    t_rec = time.perf_counter() - t0

    hr_d = torch.from_numpy(np.ascontiguousarray(hr_img, dtype=np.float32)).cuda()
    nz, nb = hr_shape[2], hr_shape[3]
    # [Z, B, X, Y] stacks of 2-D slices; spline baseline from every second HR pixel, x2 (superresDWI.py:181)
    hs = hr_d.permute(2, 3, 0, 1).contiguous()
    ss = SR_recon.permute(2, 3, 0, 1).contiguous()
    sp = baselines.rescale(hs[:, :, ::2, ::2].contiguous(), 2)
    ok = hs.amax(dim=(-2, -1)) > 0
    safe = lambda t: torch.where(ok[..., None, None], t, torch.ones_like(t)).clamp_min(1e-30)   # empty slices: finite, ignored
    ssim_spline = metrics.ssim_reference_protocol(safe(hs), safe(sp)).cpu().numpy()
    ssim_sr = metrics.ssim_reference_protocol(safe(hs), safe(ss)).cpu().numpy()
    with reports.SsimCsv(os.path.join(out_dir, "ssim_scores.csv")) as csv:
        for _slice in range(nz):
            for b in range(nb):
                csv.row(pt_id, bvalues[b] if b < len(bvalues) else b, _slice, float(ssim_spline[_slice, b]),
                        float(ssim_sr[_slice, b]))
    okn = ok.cpu().numpy()
    summary = {
        "pt_id": str(pt_id), "input": os.path.abspath(path), "lr_shape": list(lr_img.shape), "test_shape": list(test_shape),
        "n_coords": int(lr_img.size), "steps": int(args.number_of_epochs), "t_fit_s": t_fit, "t_recon_s": t_rec,
        "train_voxels_per_s": lr_img.size * args.number_of_epochs / max(t_fit, 1e-9),
        "final_loss": float(losses[-1]) if len(losses) else None,
        "psnr_db": float(metrics.psnr(hr_d, SR_recon, 1.0)),
        "psnr_spline_db": float(metrics.psnr(hs, sp, 1.0)),
        "ssim_sr_mean": float(ssim_sr[okn].mean()), "ssim_spline_mean": float(ssim_spline[okn].mean()),
    }
    rec_h, sr_h = recon.cpu().numpy(), SR_recon.cpu().numpy()
    out_vars = {"recon": rec_h, "SR_recon": sr_h, "b": np.asarray(bvalues, np.float64)}
    if maxes is not None:
        out_vars["maxes"] = maxes
    matio.savemat(os.path.join(out_dir, "recon.mat"), out_vars)
    np.save(os.path.join(out_dir, "recon.npy"), rec_h)
    with open(os.path.join(out_dir, "metrics.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print(json.dumps(summary))
    return summary


SUMMARY_KEYS = ("job", "n_coords", "steps", "t_fit_s", "t_recon_s", "train_voxels_per_s", "final_loss", "psnr_db",
                "psnr_spline_db", "ssim_sr_mean", "ssim_spline_mean")


def main(argv=None):
    """The patient loop (superresDWI.py:29).  Under ``torchrun`` (one process per GPU, WORLD_SIZE > 1) the patients are dealt
    over the ranks -- longest first by file size, the same deterministic ``dist.partition_fits`` schedule on every rank, no
    data-path collective --, every rank writes the outputs of its own patients, and ONE all_gather (RCCL) hands every rank
    the numeric summaries of all of them (rank 0 prints the list)."""
    args = build_parser().parse_args(argv)
    ids = args.pt_id if args.pt_id else [_patient_id(p) for p in args.data]
    if len(ids) != len(args.data):
        raise SystemExit("--pt_id needs one id per --data file")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return [run_patient(p, i, args) for p, i in zip(args.data, ids)]
    import torch.distributed as dist
    from mri_super_resolution_amd import dist as inr_dist
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("INR_BACKEND", "nccl"))
    rank = dist.get_rank()
    plan = inr_dist.partition_fits([float(os.path.getsize(p)) for p in args.data], world)
    local = []
    for job in plan[rank]:
        s = run_patient(args.data[job], ids[job], args)
        local.append({**{k: float(s[k]) for k in SUMMARY_KEYS if k != "job"}, "job": float(job)})
    max_jobs = max(len(p) for p in plan)
    records = sorted(inr_dist.gather_job_records(local, SUMMARY_KEYS, max_jobs), key=lambda r: r["job"])
    out = [{**r, "pt_id": str(ids[int(r["job"])]), "input": os.path.abspath(args.data[int(r["job"])]),
            "rank": next(k for k, jobs in enumerate(plan) if int(r["job"]) in jobs)} for r in records]
    if rank == 0:
        print(json.dumps({"patients": out, "world_size": world}))
    return out


if __name__ == "__main__":
    main()
