"""-m gpu: the entry points with the reference drivers' call surface (.mat in -> volume + CSV out), the spline baseline,
the acquisition products and the PerturbNet phase."""
import itertools
import json
import os

import numpy as np
import pytest
import scipy.ndimage as ndi
import torch

import mri_super_resolution_amd as inr
from mri_super_resolution_amd import baselines, contrast, drivers, matio, ops, reports
from mri_super_resolution_amd.scripts import master as master_script
from mri_super_resolution_amd.scripts import superresDWI as dwi_script
from oracle import inr_oracle as O
from oracle import torch_port as P

pytestmark = pytest.mark.gpu


def test_rescale_matches_the_scipy_call_skimage_makes():
    rng = np.random.default_rng(0)
    for shape, scale in (((25, 25), 2), ((25, 25), 4), ((60, 60), 3), ((7, 13), 2)):
        img = rng.random(shape)
        want = ndi.zoom(img, scale, order=1, mode="mirror", grid_mode=True)      # skimage 0.20 rescale(img, s) for s >= 1
        got = baselines.rescale(img, scale)
        assert got.shape == want.shape and got.dtype == np.float64
        assert np.abs(got - want).max() < 2e-7                                   # fp32 data, double weights
        assert np.abs(got - O.rescale_linear(img, scale)).max() < 2e-7
    batch = torch.rand(3, 4, 10, 12, device="cuda")
    out = baselines.rescale(batch, 2)
    assert tuple(out.shape) == (3, 4, 20, 24)
    assert np.abs(out[1, 2].cpu().numpy() - ndi.zoom(batch[1, 2].cpu().numpy().astype(np.float64), 2, order=1, mode="mirror",
                                                      grid_mode=True)).max() < 2e-7
    with pytest.raises(ValueError):
        baselines.rescale(np.ones((8, 8)), 0.5)


def test_acquisition_products_match_itertools():
    rng = np.random.default_rng(1)
    X, Y, Z, n1, n2, n3 = 3, 4, 2, 2, 3, 2
    raw = [[rng.random((X, Y, Z))], [rng.random((X, Y, Z, n1))], [rng.random((X, Y, Z, n2))], [rng.random((X, Y, Z, n3))]]
    got = drivers.acquisition_products(raw)
    assert got.shape == (X, Y, Z, 4, n1 * n2 * n3)
    for voxel in itertools.product(range(X), range(Y), range(Z)):
        want = inr.calculate_combinations(voxel, raw)                    # SRDWI.py:143-152 (host restatement)
        assert np.allclose(got[voxel], want.astype(np.float32))


def _write_volume(tmp_path, golden, pt="07"):
    vol = golden("pat07_volume.npz")["vol"]
    path = str(tmp_path / f"pat{pt}_mean_b0.mat")
    matio.savemat(path, {"data_mean_b0": vol})
    return path, vol


def test_superresDWI_entry_point_on_pat07(tmp_path, golden):
    path, vol = _write_volume(tmp_path, golden)
    out = str(tmp_path / "SR_results")
    res = dwi_script.main(["--data", path, "--output_address", out, "--number_of_epochs", "300", "--seed", "0",
                           "--roi_start", "40", "--roi_end", "90"])[0]
    d = os.path.join(out, "pat07")
    rows = reports.read_csv(os.path.join(d, "ssim_scores.csv"))
    assert open(os.path.join(d, "ssim_scores.csv")).readline() == reports.SSIM_HEADER
    assert len(rows) == 28 and set(rows[0]) == {"Pt_id", "b-value", "slice", "SSIM-spline", "SSIM-SR"}
    assert rows[5]["Pt_id"] == "07" and int(rows[5]["slice"]) == 5
    sr = np.array([float(r["SSIM-SR"]) for r in rows])
    sp = np.array([float(r["SSIM-spline"]) for r in rows])
    assert np.all((sr > 0.2) & (sr <= 1.0)) and np.all((sp > 0.2) & (sp <= 1.0))
    saved = matio.loadmat(os.path.join(d, "recon.mat"))
    assert saved["recon"].shape == (100, 100, 28, 1) and saved["SR_recon"].shape == (50, 50, 28, 1)
    assert np.array_equal(np.load(os.path.join(d, "recon.npy")), saved["recon"])
    assert saved["recon"].min() >= 0.0
    m = json.load(open(os.path.join(d, "metrics.json")))
    assert m["n_coords"] == 25 * 25 * 28 and 20.0 < m["psnr_db"] < 50.0 and m["psnr_db"] == pytest.approx(res["psnr_db"])
    # the spline column against an independent evaluation of one slice: HR/max, rescale(HR[::2, ::2], 2)/max, mask HR > 0.05
    hr = (vol / vol.max())[40:90, 40:90, 11].astype(np.float64)
    spl = ndi.zoom(hr[::2, ::2], 2, order=1, mode="mirror", grid_mode=True)
    want = O.ssim2d((hr / hr.max()) * (hr / hr.max() > 0.05), (spl / spl.max()) * (hr / hr.max() > 0.05), data_range=1.0)
    assert float(rows[11]["SSIM-spline"]) == pytest.approx(want, abs=2e-5)


def test_superresDWI_hybrid_raw_input_runs_the_perturbnet_schedule(tmp_path):
    rng = np.random.default_rng(2)
    X = Y = 32
    Z, nacq = 2, (1, 2, 2, 2)
    gx, gy = np.meshgrid(np.linspace(0, 1, X), np.linspace(0, 1, Y), indexing="ij")
    base = 100 * (1.2 + np.sin(3 * gx) * np.cos(2 * gy))
    cell = np.empty((4, 4), dtype=object)
    for b in range(4):
        for te in range(4):
            shape = (X, Y, Z) if b == 0 else (X, Y, Z, nacq[b])
            sig = base[..., None] * np.exp(-0.4 * b) * (1 - 0.1 * te)
            cell[b, te] = (sig if b == 0 else sig[..., None] * np.ones(nacq[b])) * (1 + 0.02 * rng.standard_normal(shape))
    path = str(tmp_path / "pat065_master.mat")
    matio.savemat(path, {"hybrid_raw": cell, "b": np.array([0.0, 150.0, 1000.0, 1500.0])})
    out = str(tmp_path / "res")
    res = dwi_script.main(["--data", path, "--pt_id", "65", "--output_address", out, "--number_of_epochs", "60",
                           "--pertubation_epochs", "4", "--hidden_dim", "128", "--num_layers", "2", "--mapping_size", "32",
                           "--roi_start", "2", "--roi_end", "30", "--seed", "0"])[0]
    rows = reports.read_csv(os.path.join(out, "pat65", "ssim_scores.csv"))
    assert len(rows) == Z * 4 and [float(r["b-value"]) for r in rows[:4]] == [0.0, 150.0, 1000.0, 1500.0]
    assert res["steps"] == 60 and np.isfinite(res["final_loss"])
    saved = matio.loadmat(os.path.join(out, "pat65", "recon.mat"))
    assert saved["recon"].shape == (56, 56, Z, 4) and saved["maxes"].shape == (4, 4)


def test_pn_phase_trains_perturbnet_with_the_INRmodel_flavour():
    """superresDWI.py:139-156 with a Siren that does not detach its input (INRmodel.py:147): 4 tail epochs = 2 INR steps +
    2 PerturbNet epochs over K = 3 acquisitions; trajectory against the CPU port (autograd, torch Adam)."""
    rng = np.random.default_rng(4)
    shape = (6, 5, 2)
    mean_img = rng.random(shape)
    acqs = [mean_img * (1 + 0.1 * rng.standard_normal(shape)) for _ in range(3)]
    Bm = P.fourier_matrix(3, mapping_size=16)
    B = torch.from_numpy(Bm)
    torch.manual_seed(0)
    net = inr.Siren(32, 64, 1, 1, flavor="INRmodel").cuda()
    pn = inr.PN(32, 16, 3).cuda()
    torch.manual_seed(0)
    ref = P.PortSiren(32, 64, 1, 1, flavor="INRmodel")
    ref_pn = P.PortPN(32, 16, 3)
    ds = inr.ImageFitting_set([mean_img])
    losses = drivers.fit_with_perturbnet(net, B.cuda(), ds, acqs, number_of_epochs=10, pertubation_epochs=4, PN_dim=16,
                                         lr=1e-4, perturb_lr=1e-3, perturb_net=pn)
    # CPU port of the same schedule
    x = P.port_input_mapping(P.port_mgrid(shape), B)
    tgt = torch.from_numpy(mean_img.reshape(-1, 1)).float()
    acq_t = [torch.from_numpy(a.reshape(-1, 1)).float() for a in acqs]
    opt = torch.optim.Adam(ref.parameters(), lr=1e-4)
    popt = torch.optim.Adam(ref_pn.parameters(), lr=1e-3)
    ref_losses = []
    for ctr in range(10):
        if ctr < 6 or ctr % 2:
            loss = ((ref(x) - tgt) ** 2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            ref_losses.append(loss.item())
        else:
            for k, gt in enumerate(acq_t):
                out = ref(P.port_input_mapping(ref_pn(x, k, 1 / 128.), B))
                loss = ((out - gt) ** 2).mean()
                popt.zero_grad()
                loss.backward()
                popt.step()
    assert len(losses) == 8 and np.allclose(losses, ref_losses, rtol=1e-4)
    for (n, a), b in zip(pn.named_parameters(), ref_pn.parameters()):
        assert O.rel_l2(a.detach().cpu().numpy(), b.detach().numpy()) < 1e-4, n
    # SRDWI flavour: the detached input leaves the PerturbNet untouched (SRDWI.py:88) -- the schedule still runs
    torch.manual_seed(0)
    net2 = inr.Siren(32, 64, 1, 1).cuda()
    pn2 = inr.PN(32, 16, 3).cuda()
    before = [p.detach().clone() for p in pn2.parameters()]
    drivers.fit_with_perturbnet(net2, B.cuda(), ds, acqs, number_of_epochs=10, pertubation_epochs=4, PN_dim=16, perturb_net=pn2)
    assert all(torch.equal(a, b) for a, b in zip(before, pn2.parameters()))


def test_master_entry_point_writes_the_contrast_csv(tmp_path, golden):
    """master.py on a synthetic pat07 `_alldata` (the real one was never published): 3 directions x 2 acquisitions built
    from the committed mean-b0 volume; CSV schema, row count, and the rows against a direct evaluation."""
    vol = golden("pat07_volume.npz")["vol"].astype(np.float64)
    rng = np.random.default_rng(5)
    data_dir = tmp_path / "anon_data"
    data_dir.mkdir()
    dwi = np.stack([0.4 * vol * (1 + 0.05 * rng.standard_normal(vol.shape)) for _ in range(6)], axis=-1).astype(np.float32)
    matio.savemat(str(data_dir / "pat07_alldata.mat"), {"data": dwi})
    matio.savemat(str(data_dir / "pat07_mean_b0.mat"), {"data_mean_b0": vol.astype(np.float32)})
    spec = [{"pt_id": "18-1681-07", "b": 900, "cancer_loc": [60, 70], "contralateral_loc": [60, 55], "noise": [45, 45],
             "cancer_slice": 11, "acquisitions": [2, 2, 2]}]
    with open(str(tmp_path / "cases.json"), "w") as fh:
        json.dump(spec, fh)
    out = master_script.main(["--out_folder", str(tmp_path / "exp"), "--out_img_folder", str(tmp_path / "img"),
                              "--total_steps", "40", "--seg", "10", "--hidden_layers", "2", "--hidden_features", "32",
                              "--scale", "2", "--exp_name", "t1", "--data_dir", str(data_dir),
                              "--cases", str(tmp_path / "cases.json")])
    csv_path = os.path.join(str(tmp_path / "exp"), "t1.csv")
    assert out["csv"] == csv_path and open(csv_path).readline() == reports.CONTRAST_HEADER
    rows = reports.read_csv(csv_path)
    assert len(rows) == 4 * 8 * 3                                        # (x, y, z, mean) x 8 images x (C, CNR, CNR2)
    assert {r["direction"] for r in rows} == {"x", "y", "z", "mean"} and {r["patient"] for r in rows} == {"07"}
    # the 'mean' image rows of direction x: contrast of the direction mean itself
    c = contrast.case(**spec[0], data_dir=str(data_dir))
    roi = dwi[40:100, 40:100, 11, 0:2].astype(np.float64).mean(axis=-1)
    want = contrast.calculate_contrast(c, 1, roi, 40)
    got = [float(r["performance"]) for r in rows if r["direction"] == "x" and r["image"] == "mean"]
    assert np.allclose(got, want, rtol=1e-6)
    assert all(np.isfinite(float(r["performance"])) for r in rows if r["image"] in ("mean", "ERD", "superres"))
    imgs = matio.loadmat(os.path.join(str(tmp_path / "img"), "t1", "07", "images.mat"))
    assert imgs["DWI_super"].shape == (120, 120) and imgs["ADC_mean"].shape == (60, 60)
    exp = tmp_path / "sr1_exp_3.txt"
    exp.write_text("style = directional\nsteps = 3000\nfocus = gland\nweight = False\ndepth = 2\nhidden = 64\ninput = 2\noutput = dwi")
    args = master_script.apply_experiment(master_script.build_parser().parse_args([]), master_script.read_experiment(str(exp)))
    assert (args.total_steps, args.hidden_layers, args.hidden_features, args.ROI_begin) == (3000, 2, 64, 40)
    with pytest.raises(NotImplementedError):
        contrast.save_dicom(np.zeros((2, 2)), "x.dcm")
