"""-m gpu: the driver loops (superresDWI.py / master.py restated on the fused entry points) against the CPU port."""
import numpy as np
import pytest
import torch

from mri_super_resolution_amd import drivers
from oracle import inr_oracle as O
from oracle import torch_port as P

pytestmark = pytest.mark.gpu


def test_fit_volume_cfg1_short(golden):
    """pat07 slice 11 (config 1), 60 steps: same trajectory as the reference loop on the CPU port (T3)."""
    hr = golden("pat07_slice11.npz")["hr"]
    res = drivers.fit_volume(hr, steps=60, seed=0, chunk_steps=25)
    assert res["n_coords"] == 4096 and tuple(res["recon"].shape) == (256, 256)
    # CPU port of the same protocol
    lr = np.ascontiguousarray((hr / hr.max())[::2, ::2])
    B = torch.from_numpy(drivers.fourier_matrix(2, seed=0))
    assert np.array_equal(B.numpy(), P.fourier_matrix(2))
    torch.manual_seed(0)
    ref = P.PortSiren(256, 512, 3, 1)
    x = P.port_input_mapping(P.port_mgrid(lr.shape), B)
    losses, _ = P.port_fit(ref, x, torch.from_numpy(lr.reshape(-1, 1)), 60, lr=1e-4)
    assert res["final_loss"] == pytest.approx(losses[-1], rel=2e-3)
    want = P.port_reconstruct(ref, (256, 256), B)
    assert O.rel_l2(res["recon"].cpu().numpy(), want) < 1e-3          # 60 chaotic fp32 steps; T3 bound is 50 steps
    assert 20.0 < res["psnr_db"] < 40.0 and 0.3 < res["ssim_mean"] <= 1.0
    want_psnr = O.psnr(hr / hr.max(), P.port_reconstruct(ref, (128, 128), B))
    assert res["psnr_db"] == pytest.approx(want_psnr, abs=0.05)


def test_fit_volume_3d_tiny():
    rng = np.random.default_rng(0)
    vol = rng.random((12, 10, 3)).astype(np.float32) + 0.1
    res = drivers.fit_volume(vol, steps=5, hidden_features=64, hidden_layers=1, mapping_size=16, seed=1)
    assert tuple(res["recon"].shape) == (24, 20, 3) and res["n_coords"] == 6 * 5 * 3
    assert np.isfinite(res["psnr_db"]) and np.isfinite(res["ssim_mean"]) is not None
    assert float(res["recon"].min()) >= 0.0                               # clamp(min=0), superresDWI.py:161


def test_slice_ensemble_matches_port():
    """master.py:137-160 with 3 acquisitions, 6 epochs, ensemble of the last 2, x2 grid."""
    rng = np.random.default_rng(3)
    acqs = [rng.random((20, 20)).astype(np.float32) for _ in range(3)]
    wts = [(rng.random((20, 20)) > 0.2).astype(np.float32) for _ in range(3)]
    got = drivers.fit_slice_ensemble(acqs, wts, total_steps=6, seg=2, scale=2, hidden_features=32, hidden_layers=2,
                                     lr=3e-4, seed=0)
    torch.manual_seed(0)
    net = P.PortSiren(2, 32, 2, 1)
    opt = torch.optim.Adam(lr=3e-4, params=net.parameters())
    coords = P.port_mgrid((20, 20))
    big = P.port_mgrid((40, 40))
    pred, large = np.zeros((20, 20)), np.zeros((40, 40))
    for step in range(6):
        for a, w in zip(acqs, wts):
            tgt = (2.0 * torch.from_numpy(a) - 1.0).reshape(-1, 1)
            loss = (torch.from_numpy(w).reshape(-1, 1) * (net(coords) - tgt) ** 2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
        if step >= 4:
            with torch.no_grad():
                pred += net(coords).view(20, 20).numpy()
                large += net(big).view(40, 40).numpy()
    assert got["optimizer_steps"] == 18
    assert O.rel_l2(got["predicted"], pred / 2) < 1e-4
    assert O.rel_l2(got["large"], large / 2) < 1e-4


def test_run_volumes_single_process():
    rng = np.random.default_rng(5)
    vols = [rng.random((8, 8, z)).astype(np.float32) + 0.1 for z in (3, 2, 4)]
    recs = drivers.run_volumes(vols, steps=3, hidden_features=32, hidden_layers=1, mapping_size=8)
    assert [int(r["job"]) for r in recs] == [0, 1, 2]
    assert [int(r["n_coords"]) for r in recs] == [4 * 4 * 3, 4 * 4 * 2, 4 * 4 * 4]
    assert all(np.isfinite(r["final_loss"]) for r in recs)


def test_run_volumes_concurrent_fits_keep_their_bits():
    """``run_volumes(concurrent=k)``: whole-volume fits side by side on host threads and streams of their own -- every fit ends with
    the loss, PSNR and SSIM it has when it runs alone (seeded draws are one critical section, kernels are per stream), the records
    come back in the sequential order, and a fit that fails on its thread is recorded as such."""
    rng = np.random.default_rng(5)
    vols = [rng.random((24 + 4 * (k % 3), 20, 3)).astype(np.float32) + 0.05 for k in range(6)]
    kw = dict(steps=40, chunk_steps=20)
    seq = drivers.run_volumes(vols, **kw)
    for k in (2, 3):
        con = drivers.run_volumes(vols, concurrent=k, **kw)
        assert [r["job"] for r in con] == [r["job"] for r in seq]
        for a, b in zip(seq, con):
            assert a["final_loss"] == b["final_loss"] and a["psnr_db"] == b["psnr_db"] and a["ssim_mean"] == b["ssim_mean"], (a, b)
            assert b["status"] == drivers.FIT_OK

    def flaky(volume, steps, return_recon=False, **kw2):
        if volume.shape[0] == 28:
            raise RuntimeError("boom")
        return drivers.fit_volume(volume, steps=steps, return_recon=return_recon, **kw2)

    recs = drivers.run_volumes(vols, concurrent=2, fit_fn=flaky, requeue=False, errors="record", **kw)
    assert [r["status"] for r in recs] == [drivers.FIT_ERROR if v.shape[0] == 28 else drivers.FIT_OK for v in vols]
    with pytest.raises(RuntimeError, match="boom"):           # the default: one process, nobody to take the fit over -- it raises
        drivers.run_volumes(vols, concurrent=2, fit_fn=flaky, requeue=False, **kw)


def test_fit_volume_cfg2_short(golden):
    """Config 2 input (whole pat07 volume): the first steps of the 3-D fit track the CPU port (kept short: the port
    needs seconds per step at N = 114,688)."""
    vol = golden("pat07_volume.npz")["vol"]
    assert vol.shape == (128, 128, 28)
    res = drivers.fit_volume(vol, steps=6, seed=0)
    assert res["n_coords"] == 64 * 64 * 28 and tuple(res["recon"].shape) == (256, 256, 28)
    lr = np.ascontiguousarray((vol / vol.max())[::2, ::2, :])
    B = torch.from_numpy(P.fourier_matrix(3))
    torch.manual_seed(0)
    ref = P.PortSiren(256, 512, 3, 1)
    x = P.port_input_mapping(P.port_mgrid(lr.shape), B)
    losses, _ = P.port_fit(ref, x, torch.from_numpy(lr.reshape(-1, 1)), 6, lr=1e-4)
    assert res["final_loss"] == pytest.approx(losses[-1], rel=1e-3)
    want = P.port_reconstruct(ref, (128, 128, 28), B)
    got = drivers.reconstruct(res["model"], (128, 128, 28), res["B"]).cpu().numpy()
    assert O.rel_l2(got, want) < 1e-4


def test_fit_hybrid_flow_small():
    """superresHybrid.py:57-140 on a small smooth phantom: four 4-D (x, y, z, b) fits, re-scaling, normalisation by
    the (b0, TE0) image, three-compartment fit of one slice; the hybrid-fit stage is checked against the CPU oracle
    on the very signals the driver produced."""
    from oracle import pia_oracle as PO
    from tests import pia_common as PC
    X, Y, Z = 20, 16, 3
    gx, gy = np.meshgrid(np.linspace(0, 1, X), np.linspace(0, 1, Y), indexing="ij")
    par = np.stack([0.4 + 0.2 * gx, 0.9 + 0.5 * gy, 2.8 + 0 * gx, 30 + 30 * gx, 50 + 40 * gy, 700 + 0 * gx,
                    0.2 + 0.3 * gx * gy, 0.3 + 0.2 * (1 - gx)], axis=-1)                      # [X, Y, 8]
    sig = np.stack([[PO.three_compartment(par[i, j]) for j in range(Y)] for i in range(X)])     # [X, Y, 16], b-major
    amp = 500.0 * (1.0 + 0.3 * np.sin(3 * gx) * np.cos(2 * gy))
    raw = (amp[..., None] * sig / 1000.0).reshape(X, Y, 1, 4, 4) * np.linspace(1.0, 0.9, Z).reshape(1, 1, Z, 1, 1)
    res = drivers.fit_hybrid(raw, roi=(2, 18, 2, 14), slice_index=1, steps=400, seed=0, hidden_features=128,
                             hidden_layers=2, mapping_size=32)
    assert tuple(res["recon_hybrid"].shape) == (32, 24, Z, 4, 4)
    assert res["D"].shape == res["T2"].shape == res["v"].shape == (32, 24, 3)
    assert np.allclose(res["v"].sum(axis=-1), 1.0)
    x = np.concatenate([res["D"], res["T2"], res["v"][..., :2]], axis=-1).reshape(-1, 8)
    assert np.all(x >= PO.LB) and np.all(x <= PO.UB)
    signals = res["signals"].cpu().numpy()
    assert signals.shape == (32 * 24, 16) and np.allclose(signals[:, 0], 1000.0, rtol=1e-5)
    # sanity of the INR stage: every second re-sampled voxel is close to the smooth phantom on the ROI grid (the two
    # endpoint-inclusive grids only coincide at the corners, and 400 steps of a small net are far from converged)
    rec = res["recon_hybrid"][::2, ::2, 1].cpu().numpy()
    want = raw[2:18, 2:14, 1]
    assert O.rel_l2(rec, want) < 0.1
    idx = np.arange(0, signals.shape[0], 13)
    ref = np.array([PO.trf_fit(s) for s in signals[idx]])
    PC.check_against(x[idx], ref)


def test_fit_volume_reseeds_a_fit_whose_loss_went_nan(golden):
    """SURVEY section 5 (INR_ERD.py:211-217): a diverged / collapsed fit is re-created and run again.  The first attempt's
    targets are poisoned with NaN (test hook); it stops at its first chunk, the re-seeded attempt is an ordinary fit."""
    hr = golden("pat07_slice11.npz")["hr"]
    res = drivers.fit_volume(hr, steps=40, seed=0, chunk_steps=10, _fault=lambda attempt: "nan" if attempt == 0 else None)
    assert (res["status"], res["reseeds"], res["health"]) == (drivers.FIT_RESEEDED, 1, "ok")
    assert np.isfinite(res["final_loss"]) and np.isfinite(res["psnr_db"]) and float(res["recon"].max()) > 0
    same = drivers.fit_volume(hr, steps=40, seed=7919, chunk_steps=10)          # the seed the second attempt used
    assert same["status"] == drivers.FIT_OK and same["final_loss"] == res["final_loss"]
    bad = drivers.fit_volume(hr, steps=40, seed=0, chunk_steps=10, max_reseeds=0, _fault=lambda attempt: "nan")
    assert bad["status"] == drivers.FIT_FAILED and bad["health"] == "nan" and bad["t_fit"] < same["t_fit"] * 2
    recs = drivers.run_volumes([hr, hr], steps=20, chunk_steps=10)
    assert [r["status"] for r in recs] == [drivers.FIT_OK] * 2 and all(r["requeued"] == 0.0 for r in recs)
