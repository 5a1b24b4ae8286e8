"""Callers on either side of the hot path: the reference's driver loops restated on the fused entry points.

* ``fit_volume``  -- superresDWI.py:84-162 / superresHybrid.py:79-125 for one N-D volume: max-normalise, LR =
  every second in-plane sample, Fourier features, ``Siren(2m, 512, 3, 1)``, Adam 1e-4, full-batch MSE fit, dense
  re-sampling at twice the HR in-plane grid ("x4" w.r.t. the LR grid), clamp at 0, PSNR/SSIM against the HR volume.
* ``fit_slice_ensemble`` -- master.py:130-160: small raw-coordinate SIREN on K acquisitions of one 2-D slice, one
  weighted optimizer step per acquisition per epoch, snapshot-ensemble of the last ``seg`` epochs at x1 and
  x``scale``.
* ``fit_hybrid`` -- superresHybrid.py:57-140: one 4-D (x, y, z, b) fit per echo time, re-scaling, normalisation by
  the (b = 0, TE = 0) image and the three-compartment fit of one slice (``pia.hybrid_fit_device``).
* ``run_volumes`` -- the patient loop (superresDWI.py:29) partitioned over one-process-per-GPU ranks with a final
  metric gather (``dist.py``).

Everything numerical runs on the HIP kernels; these functions only orchestrate.
"""
from __future__ import annotations

import threading
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import dist as inr_dist
from . import matio, metrics, ops
from .inr import PN, ImageFitting_set, ShardedSirenFitter, Siren, SirenFitter, input_mapping, reconstruct


def load_mat_volume(path: str, key: Optional[str] = None) -> np.ndarray:
    """Reads one array from a MATLAB v5 ``.mat`` (``anon_data/patNN_mean_b0.mat``: key ``data_mean_b0``)."""
    data = matio.loadmat(path)
    if key is None:
        keys = [k for k, v in data.items() if not k.startswith("__") and isinstance(v, np.ndarray) and v.ndim >= 2]
        if len(keys) != 1:
            raise KeyError(f"{path}: pass key=, candidates are {keys}")
        key = keys[0]
    return np.asarray(data[key])


def fourier_matrix(dim: int, mapping_size: int = 128, scale: float = 0.5, seed: Optional[int] = None) -> np.ndarray:
    """superresDWI.py:102-106: ``np.random.normal(size=(mapping_size, dim)) * scale`` as fp32 (seeded if asked)."""
    rng = np.random if seed is None else np.random.RandomState(seed)
    return (rng.normal(size=(mapping_size, dim)) * scale).astype(np.float32)


_INIT_LOCK = threading.Lock()      # torch.manual_seed + weight draws of one fit (see _fit_volume_once)
FIT_OK, FIT_RESEEDED, FIT_FAILED, FIT_ERROR = 0, 1, 2, 3      # record["status"]: see fit_volume / run_volumes


def fit_health(last_loss: float, recon: Optional[torch.Tensor] = None) -> str:
    """The reference's only recovery logic, generalised: ``INR_ERD.py:211-217`` / ``prepare_qual_images.py:176-182`` re-create
    the network when its whole output is zero (``if not model_output...max()``: a dead fit).  Here a fit is unhealthy when its
    loss is not finite ("nan": full-batch Adam can diverge) or when the clamped reconstruction is zero everywhere or not finite
    ("collapsed").  Returns "ok", "nan" or "collapsed"."""
    if last_loss is not None and not np.isfinite(last_loss):
        return "nan"
    if recon is not None and recon.numel():
        top = float(recon.max())
        if not np.isfinite(top) or not bool(torch.isfinite(recon).all()):
            return "nan"
        if top <= 0.0:
            return "collapsed"
    return "ok"


def fit_volume(volume: np.ndarray, steps: int = 2500, *, seed: Optional[int] = 0, max_reseeds: int = 1, group=None,
               _fault=None, **kwargs) -> Dict[str, object]:
    """``_fit_volume_once`` (the fit itself: arguments there) under the reference's failure rule (``fit_health``): a fit whose
    loss went non-finite or whose reconstruction collapsed to zero is run again from another seed (``seed + 7919 * attempt``: new
    Fourier matrix and new weights, as the reference re-creates its network), at most ``max_reseeds`` times.  The result carries
    ``status`` (FIT_OK, FIT_RESEEDED = healthy after a re-seed, FIT_FAILED = still unhealthy), ``health`` and ``reseeds``.
    A fit shared by a rank group re-seeds on the loss only (every member sees the same all-reduced loss and must take the same
    decision; the reconstruction exists on the group's first rank alone).  ``_fault(attempt)``: test hook (returns "nan" to
    poison that attempt's targets)."""
    attempt = 0
    while True:
        s = None if seed is None else seed + 7919 * attempt
        res = _fit_volume_once(volume, steps, seed=s, group=group, _poison=bool(_fault and _fault(attempt) == "nan"), **kwargs)
        shared = group is not None and inr_dist.is_shared(torch.distributed.get_world_size(group))
        health = fit_health(res.get("final_loss"), None if shared else res.get("_recon_probe"))
        res.pop("_recon_probe", None)
        res["health"], res["reseeds"] = health, attempt
        if health == "ok":
            res["status"] = FIT_RESEEDED if attempt else FIT_OK
            return res
        if attempt >= max_reseeds:
            res["status"] = FIT_FAILED
            return res
        attempt += 1


def _fit_volume_once(volume: np.ndarray, steps: int = 2500, hidden_features: int = 512, hidden_layers: int = 3,
                     mapping_size: int = 128, ff_scale: float = 0.5, lr: float = 1e-4, seed: Optional[int] = 0,
                     downsample: bool = True, upscale_axes: int = 2, evaluate: bool = True, chunk_steps: int = 250,
                     return_recon: bool = True, normalize: bool = True, group=None, _poison: bool = False) -> Dict[str, object]:
    """One INR super-resolution fit of an N-D volume (first ``upscale_axes`` axes are in-plane).

    ``downsample=True``: the volume is the HR ground truth, training uses ``vol[::2, ::2, ...]`` and the result is
    evaluated on the HR grid (PSNR, mean per-slice SSIM by the reference protocol) as well as re-sampled at 2x the
    HR in-plane size.  ``downsample=False``: the volume itself is the training grid and is re-sampled at 2x.
    ``group``: a process group whose ranks fit this ONE volume together, each on its contiguous share of the rows
    (``ShardedSirenFitter``); every member ends with identical weights, only the group's first rank re-samples and
    evaluates (the others return timings and the final loss).
    """
    vol = np.ascontiguousarray(volume, dtype=np.float32)
    vmax = float(vol.max()) if normalize else 1.0       # superresHybrid normalises per (b, TE) before stacking (:57-63)
    vol = np.ascontiguousarray(vol / vmax)                                                     # superresDWI.py:50-55
    sl = tuple(slice(None, None, 2) if a < upscale_axes else slice(None) for a in range(vol.ndim))
    lr_vol = np.ascontiguousarray(vol[sl]) if downsample else vol
    hr_shape = vol.shape
    test_shape = tuple(2 * s if a < upscale_axes else s for a, s in enumerate(hr_shape))
    # (seeding torch's global generator and drawing the weights from it is one critical section: `run_volumes(concurrent=k)` runs
    #  several fits of this process at once, and each must get the weights its seed gives a fit that runs alone)
    shared = group is not None and inr_dist.is_shared(torch.distributed.get_world_size(group))
    _INIT_LOCK.acquire()
    try:
        if seed is not None:
            torch.manual_seed(seed)
        B = torch.from_numpy(fourier_matrix(vol.ndim, mapping_size, ff_scale, seed)).cuda()
        if not shared:
            model = Siren(2 * mapping_size, hidden_features, hidden_layers, 1).cuda()
    finally:
        _INIT_LOCK.release()
    if shared:
        # one fit, several ranks: everybody uses the first rank's Fourier matrix (the weights follow in the fitter)
        src = torch.distributed.get_global_rank(group, 0)
        if torch.distributed.get_backend(group) == "nccl":
            torch.distributed.broadcast(B, src=src, group=group)
        else:
            Bh = B.cpu()
            torch.distributed.broadcast(Bh, src=src, group=group)
            B.copy_(Bh)
        model = Siren(2 * mapping_size, hidden_features, hidden_layers, 1).cuda()      # (the draw order of a shared fit: B, broadcast, weights)
    data = ImageFitting_set([lr_vol])
    model_input = input_mapping(data.coords[0], B)                        # built once per fit (superresDWI.py:122)
    pixels = data.pixels[0]
    if _poison:
        pixels = pixels * float("nan")
    g_size = torch.distributed.get_world_size(group) if group is not None else 1
    g_rank = torch.distributed.get_rank(group) if group is not None else 0
    torch.cuda.current_stream().synchronize()      # (this fit's stream: fits may run side by side)
    t0 = time.perf_counter()
    if shared:
        n_rows = model_input.shape[0]
        lo, hi = n_rows * g_rank // g_size, n_rows * (g_rank + 1) // g_size
        model_input, pixels = model_input[lo:hi].contiguous(), pixels[lo:hi].contiguous()
        fitter = ShardedSirenFitter(model, global_rows=n_rows, lr=lr, group=group)
    else:
        fitter = SirenFitter(model, lr=lr)
    losses = []
    done = 0
    while done < steps:
        k = min(chunk_steps, steps - done)
        losses.append(fitter.step(model_input, pixels, k))
        done += k
        if not bool(torch.isfinite(losses[-1][-1])):     # diverged: the remaining steps cannot bring it back (fit_volume re-seeds)
            break
    torch.cuda.current_stream().synchronize()
    t_fit = time.perf_counter() - t0
    fitter.release_workspace()
    if g_rank != 0:      # a partner of a sharded fit: the group's first rank owns re-sampling and evaluation
        return {"n_coords": int(lr_vol.size), "steps": int(steps), "t_fit": t_fit, "t_recon": 0.0,
                "final_loss": float(torch.cat(losses)[-1]) if steps else None,
                "first_loss": float(losses[0][0]) if steps else None, "model": model, "B": B, "partner": True}
    t0 = time.perf_counter()
    recon = reconstruct(model, test_shape, B)                            # superresDWI.py:125-126,161
    torch.cuda.current_stream().synchronize()
    t_rec = time.perf_counter() - t0
    out: Dict[str, object] = {
        "n_coords": int(lr_vol.size), "steps": int(steps), "t_fit": t_fit, "t_recon": t_rec,
        "train_voxels_per_s": lr_vol.size * steps / t_fit, "recon_voxels_per_s": recon.numel() / t_rec,
        "e2e_voxels_per_s": recon.numel() / (t_fit + t_rec), "final_loss": float(torch.cat(losses)[-1]) if steps else None,
        "first_loss": float(losses[0][0]) if steps else None, "test_shape": test_shape, "scale_back": vmax,
    }
    if evaluate and downsample:
        hr = torch.from_numpy(vol).cuda()
        sr = reconstruct(model, hr_shape, B)                             # SR_recon on the HR grid (superresDWI.py:162)
        out["psnr_db"] = float(metrics.psnr(hr, sr, 1.0))
        if vol.ndim >= 2:
            # per 2-D slice over the trailing axes, reference protocol (superresDWI.py:179-186)
            perm = tuple(range(2, vol.ndim)) + (0, 1)
            hs, ss = hr.permute(perm).contiguous(), sr.permute(perm).contiguous()
            ok = hs.amax(dim=(-2, -1)) > 0
            vals = metrics.ssim_reference_protocol(hs[ok].contiguous(), ss[ok].contiguous().clamp_min(1e-30)) \
                if vol.ndim > 2 else metrics.ssim_reference_protocol(hs, ss)
            out["ssim_mean"] = float(vals.mean())
    if return_recon:
        out["recon"] = recon
    out["_recon_probe"] = recon          # (fit_volume's health check; dropped there)
    out["model"] = model
    out["B"] = B
    return out


def acquisition_products(hybrid_raw_norm, te: int = 0) -> np.ndarray:
    """superresDWI.py:57-76 (``Pool(32).starmap(calculate_combinations, ...)`` over every voxel, each task pickling the
    whole volume) as ONE device kernel: ``hybrid_raw_norm[b][te]`` (b = 0: [X, Y, Z] or [X, Y, Z, 1]; b = 1..3:
    [X, Y, Z, n_b]) -> ``acquisitions`` [X, Y, Z, 4, n1*n2*n3] float64 on the host, combination index in
    itertools.product order (SRDWI.py:143-152)."""
    dev = ops.require_gpu()
    raws = [torch.from_numpy(np.ascontiguousarray(hybrid_raw_norm[b][te], dtype=np.float32)).to(dev) for b in range(4)]
    if raws[0].dim() == 4:
        raws[0] = raws[0][..., 0].contiguous()
    return ops.acquisition_products(*raws).cpu().numpy().astype(np.float64)


def fit_with_perturbnet(INR: Siren, B: torch.Tensor, mean_dataset: ImageFitting_set, acquisitions: Sequence[np.ndarray],
                        number_of_epochs: int = 2500, pertubation_epochs: int = 10, PN_dim: int = 128, lr: float = 1e-4,
                        perturb_lr: float = 1e-6, eps: float = 1 / 128., perturb_net: Optional[PN] = None):
    """superresDWI.py:108-156: ``number_of_epochs - pertubation_epochs`` full-batch INR steps on the mean image (one fused
    call), then the alternating tail: odd epochs one more INR step, even epochs one PerturbNet step per acquisition
    (``loss(INR(input_mapping(PN(x, sample, eps))), acquisition_sample)``, Adam lr 1e-6 on the PerturbNet only).
    All K acquisition targets are uploaded once and stay on the device (the reference copies one per step, :148).

    With the SRDWI flavour ``Siren.forward`` detaches its input (SRDWI.py:88): no gradient reaches the PerturbNet and
    ``perturb_optim.step()`` changes nothing -- kept exactly so (the loss is still evaluated).  With
    ``Siren(flavor='INRmodel')`` (INRmodel.py:147, inrDWI.py) the gradient flows INR input -> Fourier features -> PN.
    Returns the per-step INR losses (host list) and leaves the trained PerturbNet in ``fit_with_perturbnet.last_pn``."""
    model_input = input_mapping(mean_dataset.coords[0], B)
    target = mean_dataset.pixels[0]
    dataset = ImageFitting_set(list(acquisitions))                                        # K device-resident targets
    dimension = len(dataset.shape)
    pn = perturb_net if perturb_net is not None else PN(in_features=model_input.shape[1], hidden_features=PN_dim,
                                                         dimension=dimension).cuda()
    fitter = SirenFitter(INR, lr=lr)
    head = max(number_of_epochs - pertubation_epochs, 0)
    losses = [fitter.step(model_input, target, head)] if head else []
    pn_params = [p for p in pn.parameters()]
    pn_state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in pn_params]
    pn_steps, pn_losses = 0, []
    for ctr in range(head, number_of_epochs):
        if ctr % 2:
            losses.append(fitter.step(model_input, target, 1))
            continue
        for sample in range(len(dataset)):
            ground_truth = dataset.pixels[sample]
            perturbed_input = input_mapping(pn(model_input, sample, eps), B)
            model_output = INR(perturbed_input)
            loss = ((model_output - ground_truth) ** 2).mean()
            for p in pn_params:
                p.grad = None
            loss.backward()
            pn_losses.append(loss.detach())
            if all(p.grad is not None for p in pn_params):                                 # INRmodel flavour only
                pn_steps += 1
                for p, (m, v) in zip(pn_params, pn_state):
                    ops.adam_step(p.data, p.grad.contiguous(), m, v, pn_steps, perturb_lr)
            for p in INR.parameters():                                                    # inr_optim.zero_grad() of the next
                p.grad = None                                                             # INR step clears these (:136)
    fit_with_perturbnet.last_pn = pn
    fit_with_perturbnet.last_pn_losses = [float(l) for l in pn_losses]
    return [float(v) for v in torch.cat(losses).cpu()] if losses else []


def fit_slice_ensemble(acquisitions: Sequence[np.ndarray], weights: Optional[Sequence[np.ndarray]] = None,
                       total_steps: int = 3000, seg: int = 150, scale: int = 3, hidden_features: int = 64,
                       hidden_layers: int = 6, lr: float = 3e-4, seed: Optional[int] = 0,
                       divide_by: Optional[int] = None) -> Dict[str, object]:
    """master.py:130-160 for one gradient direction: raw 2-D coordinates -> Siren(2, H, L, 1); per epoch one weighted
    Adam step per acquisition (targets are ``2*img-1``, nn_mri.py:174-180); the outputs of the last ``seg`` epochs at
    the native and the x``scale`` grid are averaged.  Returns float64 host arrays like the reference."""
    imgs = [np.asarray(a, np.float32) for a in acquisitions]
    side = imgs[0].shape[0]
    if any(a.shape != (side, side) for a in imgs):
        raise ValueError("acquisitions must be square 2-D images of one size")
    if seed is not None:
        torch.manual_seed(seed)
    model = Siren(2, hidden_features, hidden_layers, 1).cuda()
    coords = ImageFitting_set([imgs[0]]).coords[0]                        # get_mgrid(side, 2)
    # all acquisitions resident as ONE [K, N] tensor: a whole epoch (one weighted Adam step per acquisition) -- and every
    # epoch before the averaging window -- is a single call (inr_siren_fit_cycle), not 2 K launches per epoch
    targets = torch.stack([(2.0 * torch.from_numpy(a) - 1.0).reshape(-1) for a in imgs]).cuda()
    wts = None if weights is None else torch.stack([torch.from_numpy(np.asarray(w, np.float32)).reshape(-1)
                                                    for w in weights]).cuda()
    n_acq = len(imgs)
    fitter = SirenFitter(model, lr=lr)
    predicted = torch.zeros(side, side, dtype=torch.float64, device="cuda")
    large = torch.zeros(side * scale, side * scale, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    quiet = max(0, total_steps - seg)                                     # epochs before the first snapshot
    if quiet:
        fitter.step_cycle(coords, targets, quiet * n_acq, wts)
    for step in range(quiet, total_steps):
        fitter.step_cycle(coords, targets, n_acq, wts)
        if step >= total_steps - seg:                                     # master.py:149-160
            predicted += reconstruct(model, (side, side), None, clamp_min=None).double()
            large += reconstruct(model, (side * scale, side * scale), None, clamp_min=None).double()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # master.py:162-164 divides by args.seg whatever the number of snapshots taken (step >= total - seg gives seg of them
    # when total_steps >= seg); `divide_by` reproduces that literally
    n_snap = divide_by if divide_by is not None else min(seg, total_steps)
    return {"predicted": (predicted / max(n_snap, 1)).cpu().numpy(), "large": (large / max(n_snap, 1)).cpu().numpy(),
            "seconds": dt, "optimizer_steps": total_steps * n_acq,
            "train_voxels_per_s": total_steps * n_acq * side * side / dt, "model": model}


RECORD_KEYS = ("job", "n_coords", "steps", "t_fit", "t_recon", "psnr_db", "ssim_mean", "final_loss", "status", "reseeds", "rank",
               "requeued", "first_loss")      # first_loss: the loss of the step-0 weights (a fit's own yardstick for "it went down")


def hybrid_te_groups(world_size: int) -> List[List[int]]:
    """Which ranks fit which of the four echo times (superresHybrid.py:79): with four or more ranks every TE gets its own
    contiguous group (8 ranks: pairs, each pair row-shards its fit); with fewer, TE ``t`` goes to rank ``t % world``."""
    if world_size >= 4:
        return inr_dist._split_ranks(world_size, [1.0] * 4)
    return [[te % max(world_size, 1)] for te in range(4)]


def _sum_over_ranks(t: torch.Tensor) -> None:
    if torch.distributed.get_backend() == "nccl":
        torch.distributed.all_reduce(t)
    else:
        h = t.cpu()
        torch.distributed.all_reduce(h)
        t.copy_(h)


def fit_hybrid(hybrid_raw: np.ndarray, roi: Optional[Sequence[int]] = None, slice_index: Optional[int] = None,
               steps: int = 2500, seed: Optional[int] = 0, distributed: bool = False, gather_recon: bool = False,
               target_dtype=None, **fit_kwargs) -> Dict[str, object]:
    """superresHybrid.py:57-140.  ``hybrid_raw``: [X, Y, Z, 4 (b), 4 (TE)].  Per echo time one 4-D INR fit of the
    ROI (LR = every second in-plane voxel, coordinates (x, y, z, b)), re-sampled at twice the ROI size; the 16
    re-scaled images are normalised by the (b=0, TE=0) one (x1000) and one z-slice goes through the three-compartment
    fit.  Returns the device tensor ``recon_hybrid`` [2sx, 2sy, Z, 4, 4] and numpy maps ``D``, ``T2``, ``v``
    [2sx, 2sy, 3] plus timings.

    ``distributed=True`` (inside an initialised process group): the four TE fits are independent, so they are placed on the
    ranks by ``hybrid_te_groups`` (a group of several ranks row-shards its fit); the only exchange is ONE all-reduce of
    the z-slice that feeds the three-compartment fit ([2sx, 2sy, 4, 4]; every rank contributes the echo times it owns,
    zeros elsewhere), after which every rank runs the same hybrid fit.  ``recon_hybrid`` holds the owned echo times only
    unless ``gather_recon`` asks for a second all-reduce of the whole tensor.
    ``target_dtype=np.float16``: the normalised volume is held in half precision (BASELINE config 5) and widened per fit."""
    from . import pia
    raw = np.asarray(hybrid_raw, dtype=np.float32)
    if raw.ndim != 5 or raw.shape[3] != 4 or raw.shape[4] != 4:
        raise ValueError(f"hybrid_raw must be [X, Y, Z, 4, 4], got {raw.shape}")
    x0, x1, y0, y1 = roi if roi is not None else (0, raw.shape[0], 0, raw.shape[1])
    maxes = raw.reshape(-1, 4, 4).max(axis=0)                                              # superresHybrid.py:57-63
    norm = raw / maxes
    sx, sy, nz = x1 - x0, y1 - y0, raw.shape[2]
    recon_hybrid = torch.empty((2 * sx, 2 * sy, nz, 4, 4), dtype=torch.float32, device="cuda")
    scale = torch.from_numpy(maxes).cuda()
    t_fit = t_recon = 0.0
    losses = []
    if target_dtype is not None:
        norm = norm.astype(target_dtype)
    spread = distributed and torch.distributed.is_initialized() and inr_dist.is_shared(torch.distributed.get_world_size())
    rank = torch.distributed.get_rank() if spread else 0
    te_ranks = hybrid_te_groups(torch.distributed.get_world_size()) if spread else [[0]] * 4
    # (group creation is collective over the default group: every rank asks for every group, in the same order; cached)
    te_groups = [inr_dist.rank_group(r) if spread else None for r in te_ranks]
    if spread:
        recon_hybrid.zero_()
    owned = []
    for te in range(4):                                                                    # superresHybrid.py:79
        if rank not in te_ranks[te]:
            losses.append(float("nan"))
            continue
        vol = np.ascontiguousarray(norm[x0:x1, y0:y1, :, :, te], dtype=np.float32)         # (x, y, z, b)
        res = fit_volume(vol, steps=steps, seed=None if seed is None else seed + te, evaluate=False, normalize=False,
                         group=te_groups[te], **fit_kwargs)
        t_fit += res["t_fit"]
        t_recon += res["t_recon"]
        losses.append(res["final_loss"])
        if not res.get("partner"):
            recon_hybrid[..., te] = res["recon"] * scale[:, te]                            # superresHybrid.py:121-122
            owned.append(te)
    k = nz // 2 if slice_index is None else int(slice_index)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if spread:
        if gather_recon:
            _sum_over_ranks(recon_hybrid)
            sl = recon_hybrid[:, :, k].double()
        else:
            plane = recon_hybrid[:, :, k].contiguous()                                     # owned echo times, zeros elsewhere
            _sum_over_ranks(plane)
            sl = plane.double()
    else:
        sl = recon_hybrid[:, :, k].double()                                                # [2sx, 2sy, 4, 4]
    signals = (1000.0 * sl / (sl[..., 0:1, 0:1] + 1e-7)).reshape(-1, 16)                   # superresHybrid.py:131-138
    fit = pia.hybrid_fit_device(signals)
    x = fit["params"].cpu().numpy()
    t_hybrid = time.perf_counter() - t0
    shape = (2 * sx, 2 * sy, 3)
    v = np.concatenate([x[:, 6:8], 1 - x[:, 6:7] - x[:, 7:8]], axis=1)
    return {"recon_hybrid": recon_hybrid, "signals": signals, "D": x[:, 0:3].reshape(shape), "T2": x[:, 3:6].reshape(shape),
            "v": v.reshape(shape), "status": fit["status"].cpu().numpy().reshape(shape[:2]), "t_fit": t_fit,
            "t_recon": t_recon, "t_hybrid_fit": t_hybrid, "final_losses": losses, "slice_index": k, "owned_te": owned}


def plan_volumes(volumes: Sequence[np.ndarray], steps: int, world: int, allow_sharding: bool = True, **fit_kwargs):
    """The schedule ``run_volumes`` follows (same on every rank, no communication).  For the reference's network
    (Siren(256, 512, 3, 1): the defaults of ``fit_volume``) the planner prices a job with the MEASURED step-time table of
    ``dist.StepTimeModel`` -- a step is not linear in the rows, which is what decides whether row-sharding pays -- otherwise
    with rows x steps."""
    rows = [float(np.prod([-(-s // 2) if a < 2 else s for a, s in enumerate(v.shape)])) for v in volumes]
    costs = [r * steps for r in rows]
    # priced by the measured step-time table for the reference's network, by the linear model derived from it for any other shape
    # (dist.StepTimeModel.for_network); a job list that does not train on the half-resolution grid keeps rows x steps
    shard_time = None
    if fit_kwargs.get("downsample", True):
        model = inr_dist.StepTimeModel.for_network(2 * fit_kwargs.get("mapping_size", 128), fit_kwargs.get("hidden_features", 512),
                                                   fit_kwargs.get("hidden_layers", 3))
        shard_time = lambda j, k: model.fit_seconds(rows[j], steps, k)     # noqa: E731
    if allow_sharding and world > 1:
        plan = inr_dist.plan_fits(costs, world, shard_time=shard_time)
    else:
        whole = inr_dist.partition_fits([shard_time(j, 1) for j in range(len(rows))] if shard_time else costs, world)
        t = (lambda j: shard_time(j, 1)) if shard_time else (lambda j: costs[j])
        loads = [sum(t(j) for j in jobs) for jobs in whole]
        plan = {"gangs": [], "whole": whole, "makespan": max(loads, default=0.0), "loads": loads}
    plan["one_rank"] = sum(shard_time(j, 1) for j in range(len(rows))) if shard_time else sum(costs)
    plan["unit"] = "seconds (measured step-time table, modelled all-reduce)" if shard_time else "rows x steps"
    return plan


class FitError(RuntimeError):
    """``run_volumes``: fits that raised on their rank and that no surviving rank could take over (``.records`` = all records)."""

    def __init__(self, message, records):
        super().__init__(message)
        self.records = records


def run_volumes(volumes: Sequence[np.ndarray], steps: int = 2500, allow_sharding: bool = True, stats: Optional[dict] = None,
                fit_fn=None, requeue: bool = True, concurrent: int = 1, errors: str = "raise", plan: Optional[dict] = None,
                **fit_kwargs) -> List[Dict[str, float]]:
    """Fits every volume once over the ranks of the current process group and returns the gathered per-fit metric
    records on every rank (one RCCL all_gather).  Schedule: ``plan_volumes`` / ``dist.plan_fits`` -- the volumes that do not
    fill a whole round are fitted first, each row-sharded over its own group of ranks, the rest are packed whole (LPT);
    ``allow_sharding=False`` is plain LPT packing (no collective on the data path at all).  ``stats`` (a dict) receives the
    plan and this rank's busy seconds.  ``plan``: a schedule of ``dist.plan_fits`` form (``{"gangs": [(job, ranks), ...], "whole": [[job,
    ...] per rank]}``) to follow instead of the one ``plan_volumes`` would price -- every rank must pass the same one.

    Failure handling (the job list is idempotent -- one record per fit).  Every record carries ``status``:
    FIT_OK, FIT_RESEEDED (the fit diverged or collapsed and a re-seeded run was healthy: ``fit_volume``), FIT_FAILED (unhealthy
    after the re-seeds: kept, marked, metrics as measured) or FIT_ERROR (the fit RAISED on its rank -- a device error, an
    allocation failure).  With ``requeue`` the whole-volume fits that ended in FIT_ERROR are dealt, longest first, to the ranks
    that reported no error (the survivors; every rank derives the same assignment from the gathered records), run there and
    gathered once more; ``requeued`` = 1 marks their records.  Only RUNTIME failures are turned into records (``RuntimeError`` and
    its subclasses: device errors, ``InrHipError``, ``torch.cuda.OutOfMemoryError``); a programming error (``TypeError`` from a bad
    keyword, ``ValueError``, ...) propagates at once, as does any failure when this process is the whole job (world 1: nobody could
    take the fit over, the first exception is re-raised after the loop).  ``errors="raise"`` (default): records still FIT_ERROR after
    the re-queue raise ``FitError`` on every rank (all ranks hold the same gathered records, so all raise); ``errors="record"`` returns
    them, NaN metrics and all.  A fit shared by a rank group is not re-queued: a member that
    raises mid-fit leaves its partners in the gradient all-reduce, which this layer cannot repair -- the exception propagates.
    ``fit_fn(volume, steps=..., return_recon=False, **fit_kwargs)`` replaces ``fit_volume`` (tests of the scheduling itself).

    ``concurrent`` (default 1): whole-volume fits a rank runs side by side, each on a host thread and HIP stream of its own.  A fit
    of a few thousand rows keeps a chip busy a fifth of the time (one 64 x 128 tile per CU and launch, 13 launches per step): two at
    once measured 1.41 x the aggregate rate at 4,096 rows, 1.26 x at 16,384 (``tools/concurrent_small.py``); volumes that fill the
    chip gain nothing.  Every fit keeps the bits it has alone (seeded draws are one critical section, kernels are per stream)."""
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    fit = fit_fn or fit_volume
    if plan is None:
        plan = plan_volumes(volumes, steps, world, allow_sharding, **fit_kwargs)
    else:
        jobs = sorted([j for j, _ in plan["gangs"]] + [j for w in plan["whole"] for j in w])
        if jobs != list(range(len(volumes))) or len(plan["whole"]) != world:
            raise ValueError("plan must place every volume exactly once and list the whole-volume jobs of every rank")
    local = []
    t_start = time.perf_counter()
    nan = float("nan")

    def record(job, res, requeued=0.0):
        local.append({"job": job, "n_coords": res["n_coords"], "steps": steps, "t_fit": res["t_fit"],
                      "t_recon": res["t_recon"], "psnr_db": res.get("psnr_db", nan),
                      "ssim_mean": res.get("ssim_mean", nan),
                      "final_loss": nan if res.get("final_loss") is None else res["final_loss"],
                      "first_loss": nan if res.get("first_loss") is None else res["first_loss"],
                      "status": float(res.get("status", FIT_OK)), "reseeds": float(res.get("reseeds", 0)), "rank": float(rank),
                      "requeued": requeued})

    if errors not in ("raise", "record"):
        raise ValueError(f"errors must be 'raise' or 'record', got {errors!r}")
    raised = []

    def run_whole(job, requeued=0.0):
        try:
            record(job, fit(volumes[job], steps=steps, return_recon=False, **fit_kwargs), requeued)
        except RuntimeError as e:      # device / library / allocation failures; the record says so, the survivors take the job over
            import sys
            raised.append(e)
            print(f"[run_volumes] rank {rank}: fit of volume {job} raised {type(e).__name__}: {e}", file=sys.stderr)
            record(job, {"n_coords": float(np.asarray(volumes[job]).size), "t_fit": nan, "t_recon": nan, "final_loss": nan,
                         "status": FIT_ERROR}, requeued)

    # gang phase: every rank creates every group (new_group is collective over the default group), then works in its own
    groups = [(job, ranks, inr_dist.rank_group(ranks)) for job, ranks in plan["gangs"]]
    for job, ranks, grp in groups:
        if rank in ranks:
            res = fit(volumes[job], steps=steps, return_recon=False, group=grp, **fit_kwargs)
            if not res.get("partner"):
                record(job, res)
    whole = list(plan["whole"][rank])
    # (networks small enough for the persistent cooperative kernel -- grid barriers over co-resident blocks: siren_small.hip -- are
    #  never run side by side: two such launches could each hold part of the chip and wait for the rest)
    cooperative = fit_kwargs.get("hidden_features", 512) in (32, 64) and 2 * fit_kwargs.get("mapping_size", 128) <= 32
    if int(concurrent) > 1 and len(whole) > 1 and torch.cuda.is_available() and not cooperative:
        from concurrent.futures import ThreadPoolExecutor
        first = len(local)
        tls = threading.local()

        def on_own_stream(job):
            if not hasattr(tls, "stream"):
                tls.stream = torch.cuda.Stream()
            with torch.cuda.stream(tls.stream):
                run_whole(job)
                tls.stream.synchronize()

        torch.cuda.synchronize()                           # (the side streams start behind whatever the caller enqueued)
        with ThreadPoolExecutor(max_workers=min(int(concurrent), len(whole))) as pool:
            list(pool.map(on_own_stream, whole))
        local[first:] = sorted(local[first:], key=lambda r: whole.index(r["job"]))      # (records in the order of the sequential run)
    else:
        for job in whole:
            run_whole(job)
    if stats is not None:
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        stats["busy_s"] = time.perf_counter() - t_start
        stats["plan"] = plan
    max_jobs = max((len(p) for p in plan["whole"]), default=0) + 1
    records = inr_dist.gather_job_records(local, RECORD_KEYS, max_jobs)
    failed = sorted((r for r in records if r["status"] == FIT_ERROR), key=lambda r: (-r["n_coords"], r["job"]))
    if requeue and failed:
        bad_ranks = {int(r["rank"]) for r in failed}
        survivors = [k for k in range(world) if k not in bad_ranks]
        if survivors:
            busy = {k: sum(r["t_fit"] for r in records if int(r["rank"]) == k and r["t_fit"] == r["t_fit"]) for k in survivors}
            mine = []
            for r in failed:                               # longest first, each to the survivor that is free first
                k = min(survivors, key=lambda q: (busy[q], q))
                busy[k] += r["n_coords"] * steps / 5.5e7      # (seconds at the fused step's ~55 M coordinate-steps/s)
                if k == rank:
                    mine.append(int(r["job"]))
            local.clear()
            for job in mine:
                run_whole(job, requeued=1.0)
            second = inr_dist.gather_job_records(local, RECORD_KEYS, len(failed))
            redone = {int(r["job"]): r for r in second}
            records = [redone.get(int(r["job"]), r) if r["status"] == FIT_ERROR else r for r in records]
            if stats is not None:
                stats["requeued"] = sorted(redone)
    records = sorted(records, key=lambda r: r["job"])
    still = [int(r["job"]) for r in records if r["status"] == FIT_ERROR]
    if still and errors == "raise":
        if world == 1 and raised:
            raise raised[0]                                # single process: the caller sees the exception itself
        raise FitError(f"run_volumes: fits of volumes {still} raised and no rank could take them over "
                       f"(errors='record' returns the NaN records instead)", records)
    return records
