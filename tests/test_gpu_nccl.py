"""-m gpu: the RCCL branches of the multi-GPU path, executed on the ONE test GPU.

Every collective of this package is skipped when its group has one rank, and the CPU / rehearsal tests take the `gloo` arm
(host-staged, synchronous), so until round 5 no `backend == "nccl"` line had run anywhere.  Here a child process initialises a
world-size-1 `nccl` group (RCCL; `device_id` bound) and `INR_FORCE_COLLECTIVES=1` (`dist.force_collectives`) sends every
collective through: a sum / broadcast / gather over one rank is the identity, so each result must equal, BIT FOR BIT, the run
without the collective.  That is the check available here that the library's launches (enqueued on torch's current HIP stream)
and RCCL's collective (on its own stream, joined to the current one by events) are ordered correctly: a collective that
overtook `inr_siren_loss_grad`, or an Adam step that overtook the collective, would read a half-written gradient buffer.
Reference: superresDWI.py:29,132-138 (the loop being sharded), SURVEY.md section 8(e)."""
import os

import numpy as np
import pytest
import torch

from tests.mp_util import run_ranks

pytestmark = pytest.mark.gpu


def _problem(n=5000, seed=2):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy((rng.random((n, 64)) * 2 - 1).astype(np.float32))
    t = torch.from_numpy(rng.random((n, 1)).astype(np.float32))
    return x, t


def _sharded_run(force, steps, stream=None):
    import torch.distributed as dist

    import mri_super_resolution_amd as inr
    os.environ["INR_FORCE_COLLECTIVES"] = "1" if force else "0"
    x, t = _problem()
    torch.manual_seed(0)
    net = inr.Siren(64, 128, 2, 1).cuda()
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        xs, ts = x.cuda(), t.cuda()
        fitter = inr.ShardedSirenFitter(net, global_rows=x.shape[0], lr=1e-4, group=dist.group.WORLD)
        losses = [fitter.step(xs, ts, 1) for _ in range(steps // 2)]        # single steps: collective between every pair of launches
        losses.append(fitter.step(xs, ts, steps - steps // 2))              # and one multi-step call
        torch.cuda.current_stream().synchronize()
        return torch.cat(losses).cpu().numpy(), fitter.flat.cpu().numpy(), fitter.step_count


def _nccl_worker(rank, world):
    import torch.distributed as dist

    from mri_super_resolution_amd import dist as inr_dist
    from mri_super_resolution_amd import drivers
    assert dist.get_backend() == "nccl" and world == 1
    out = {}
    # -- ShardedSirenFitter: broadcast of weights / moments / step count at construction, all-reduce of [gradient | loss] per step
    plain = _sharded_run(False, 24)
    assert not inr_dist.force_collectives() or True
    forced = _sharded_run(True, 24)
    side = _sharded_run(True, 24, stream=torch.cuda.Stream())               # the same on a non-default current stream
    out["fitter"] = (plain, forced, side)
    # -- record gathers (all_gather_into_tensor on device tensors)
    os.environ["INR_FORCE_COLLECTIVES"] = "1"
    assert inr_dist.force_collectives()
    rec = {"rank": 0.0, "n": 524288.0, "seconds": 1.25, "final_loss": 3.5e-4}
    out["gather_records"] = (inr_dist.gather_records(rec), rec)
    local = [{"job": 3.0, "a": 1.5, "b": float("nan")}, {"job": 7.0, "a": -2.0, "b": 4.0}]
    out["gather_job_records"] = (inr_dist.gather_job_records(local, ("job", "a", "b"), 4), local)
    # -- fit_volume(group=...): the Fourier matrix broadcast + the sharded fitter inside the driver
    gx, gy = np.meshgrid(np.linspace(0, 1, 24), np.linspace(0, 1, 20), indexing="ij")
    vol = np.stack([(0.4 + 0.3 * np.sin(4 * gx) * np.cos(3 * gy)) * (1 + 0.1 * k) for k in range(3)], axis=-1).astype(np.float32)
    kw = dict(steps=30, hidden_features=64, hidden_layers=1, mapping_size=16, seed=0, chunk_steps=10, return_recon=False)
    shared = drivers.fit_volume(vol, group=dist.group.WORLD, **kw)
    os.environ["INR_FORCE_COLLECTIVES"] = "0"
    alone = drivers.fit_volume(vol, group=dist.group.WORLD, **kw)           # group of one, no hook: the plain fused fit
    os.environ["INR_FORCE_COLLECTIVES"] = "1"
    out["fit_volume"] = {k: (shared[k], alone[k]) for k in ("final_loss", "first_loss", "psnr_db", "n_coords")}
    # -- run_volumes: its gather (and, with the hook, nothing else: one rank has no gangs)
    recs = drivers.run_volumes([vol, vol[:, :, :2]], steps=10, hidden_features=64, hidden_layers=1, mapping_size=16, seed=0, chunk_steps=10)
    out["run_volumes"] = recs
    # -- fit_hybrid(distributed=True): every TE fit through the shared path, then the all-reduce of the fitted slice
    X, Y, Z = 20, 16, 3
    hx, hy = np.meshgrid(np.linspace(0, 1, X), np.linspace(0, 1, Y), indexing="ij")
    amp = 500.0 * (1.0 + 0.3 * np.sin(3 * hx) * np.cos(2 * hy))
    decay = np.exp(-np.arange(4)[:, None] * 0.35 - np.arange(4)[None, :] * 0.25)
    raw = (amp[:, :, None, None, None] * decay * np.linspace(1.0, 0.9, Z).reshape(1, 1, Z, 1, 1)).astype(np.float32)
    hkw = dict(roi=(2, 18, 2, 14), slice_index=1, steps=40, seed=0, hidden_features=64, hidden_layers=1, mapping_size=16)
    h_forced = drivers.fit_hybrid(raw, distributed=True, **hkw)
    os.environ["INR_FORCE_COLLECTIVES"] = "0"
    h_plain = drivers.fit_hybrid(raw, distributed=True, **hkw)
    out["hybrid"] = (h_forced["signals"].cpu().numpy(), h_plain["signals"].cpu().numpy(), h_forced["owned_te"])
    return out


def test_rccl_branches_on_a_one_rank_group():
    out, = run_ranks(_nccl_worker, 1, timeout=600, backend="nccl")
    (pl, pf, pc), (fl, ff, fc), (sl, sf, sc) = out["fitter"]
    assert pc == fc == sc == 24
    assert np.array_equal(pl, fl) and np.array_equal(pf, ff), "all-reduce / broadcast over one rank must be the identity, bit for bit"
    assert np.array_equal(pl, sl) and np.array_equal(pf, sf), "same on a non-default current stream"
    assert np.isfinite(pl).all() and pl[-1] < pl[0]
    got, rec = out["gather_records"]
    assert got == [rec]
    got, local = out["gather_job_records"]
    assert len(got) == 2 and all(g["job"] == l["job"] and g["a"] == l["a"] for g, l in zip(got, local))
    assert got[0]["b"] != got[0]["b"] and got[1]["b"] == 4.0
    fv = out["fit_volume"]
    assert fv["n_coords"][0] == fv["n_coords"][1]
    assert fv["first_loss"][0] == pytest.approx(fv["first_loss"][1], rel=1e-5)      # same draws (B, weights): same starting point
    assert fv["final_loss"][0] == pytest.approx(fv["final_loss"][1], rel=2e-3)      # sharded step vs fused step: other summation order
    assert fv["psnr_db"][0] == pytest.approx(fv["psnr_db"][1], abs=0.05)
    recs = out["run_volumes"]
    assert [int(r["job"]) for r in recs] == [0, 1] and all(np.isfinite(r["final_loss"]) and r["status"] == 0.0 for r in recs)
    hs, hp, owned = out["hybrid"]
    assert owned == [0, 1, 2, 3]
    assert np.abs(hs - hp).max() / np.abs(hp).max() < 5e-3
