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
from mri_super_resolution_amd.scripts import rams_master as rams_master_script
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


def test_rams_master_entry_point_on_a_synthetic_alldata_file(tmp_path):
    """multi-image-super-resolution/master.py:20-62 end to end: `patNN_alldata.mat` (128, 128, Z, T) + `patNN_mean_b0.mat` written
    by matio -> x256 -> random 9-subsets through RAMS(3, 32, 3, 9, 8, 12) -> mean (384, 384) -> ADC -> .npy / .mat.  The subset
    forwards are checked against the host restatement (oracle/rams_port.py: parity unpinned against TensorFlow), the
    ADC against the reference's formula evaluated here."""
    import random

    from mri_super_resolution_amd import rams
    from oracle import rams_port as R
    rng = np.random.default_rng(8)
    gx, gy = np.meshgrid(np.linspace(0, 1, 128), np.linspace(0, 1, 128), indexing="ij")
    base = 60.0 + 50.0 * np.sin(5 * gx) * np.cos(4 * gy)
    Z, T = 3, 12
    dwi = np.stack([np.stack([base * (1 + 0.1 * z) * (1 + 0.04 * rng.standard_normal(base.shape)) for _ in range(T)], axis=-1)
                    for z in range(Z)], axis=2).clip(1, 250).astype(np.float32)                      # [128, 128, Z, T]
    b0 = np.stack([2.5 * base * (1 + 0.1 * z) for z in range(Z)], axis=2).astype(np.float32)
    data_dir = tmp_path / "anon_data"
    data_dir.mkdir()
    matio.savemat(str(data_dir / "pat09_alldata.mat"), {"data": dwi})
    matio.savemat(str(data_dir / "pat09_mean_b0.mat"), {"data_mean_b0": b0})
    spec = [{"pt_id": "18-1681-09", "b": 900, "cancer_loc": [60, 70], "contralateral_loc": [60, 55], "noise": [45, 45],
             "cancer_slice": 1, "acquisitions": [4, 4, 4]}]
    with open(str(tmp_path / "cases.json"), "w") as fh:
        json.dump(spec, fh)
    params = R.init_rams_params(seed=4, perturb_g=True)
    weights = rams.RAMS(3, 32, 3, 9, 8, 12, params=params).save_weights(str(tmp_path / "rams.npz"))
    out = rams_master_script.main(["--out_folder", str(tmp_path / "exp"), "--out_img_folder", str(tmp_path / "img"), "--exp_name", "mi1",
                                   "--data_dir", str(data_dir), "--cases", str(tmp_path / "cases.json"), "--weights", weights,
                                   "--sample_size", "2", "--seed", "3"])
    rec, = out["cases"]
    assert rec["patient"] == "09" and rec["shape"] == [384, 384] and rec["sample_size"] == 2
    draws = random.Random(3)
    assert rec["subsets"] == [draws.sample(list(range(T)), 9) for _ in range(2)]                  # master.py:46 under --seed
    mean = np.load(os.path.join(rec["out_dir"], "DWI_mean.npy"))
    adc = np.load(os.path.join(rec["out_dir"], "ADC_mean.npy"))
    mat = matio.loadmat(os.path.join(rec["out_dir"], "images.mat"))
    assert mean.shape == adc.shape == (384, 384) and np.array_equal(mat["DWI_mean"], mean) and np.array_equal(mat["ADC_mean"], adc)
    lor = dwi[:, :, 1, :][None].astype("uint16") * 256                                            # master.py:40-42
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    want = np.mean([R.predict_tensor(params, lor[:, :, :, s].astype(np.float32))[0, :, :, 0] for s in rec["subsets"]], axis=0)
    assert np.abs(mean - want).max() <= 1.0 and float((mean != want).mean()) < 4e-2              # rounded outputs: half-integer flips only
    b0_scaled = ndi.zoom(b0[:, :, 1].astype(np.float64), 3, order=1, mode="mirror", grid_mode=True)
    want_adc = -np.log((mean / (b0_scaled + 1e-7)) + 1e-7) / 900 * 1000000                       # master.py:53-57
    assert np.allclose(adc, want_adc, rtol=1e-5, atol=1e-3)
    assert os.path.exists(os.path.join(str(tmp_path / "exp"), "mi1.json"))
    with pytest.raises(SystemExit):                                                                  # no patient table: says so
        rams_master_script.main(["--out_folder", str(tmp_path / "exp"), "--data_dir", str(data_dir)])


def _hybrid_master(path, X=32, Z=3, seed=2):
    """A synthetic master.mat in the reference's layout: hybrid_raw [b][TE] cell, b = 0: [X, Y, Z], b > 0: [X, Y, Z, acq]."""
    from oracle import pia_oracle as PO
    rng = np.random.default_rng(seed)
    gx, gy = np.meshgrid(np.linspace(0, 1, X), np.linspace(0, 1, X), indexing="ij")
    v_ep = 0.25 + 0.2 * np.sin(3 * gx) * np.cos(2 * gy)
    v_st = 0.45 - 0.1 * gx
    p = np.stack([np.full_like(gx, 0.5), np.full_like(gx, 1.2), np.full_like(gx, 2.9), np.full_like(gx, 50.0),
                  np.full_like(gx, 80.0), np.full_like(gx, 700.0), v_ep, v_st], axis=-1).reshape(-1, 8)
    sig = np.stack([PO.three_compartment(q, PO.B16, PO.TE16) for q in p]).reshape(X, X, 4, 4) / 1000.0     # [x, y, b, te]
    nacq = (1, 2, 2, 2)
    cell = np.empty((4, 4), dtype=object)
    for b in range(4):
        for te in range(4):
            base = 200.0 * sig[:, :, b, te][..., None] * np.ones(Z)
            shape = (X, X, Z) if b == 0 else (X, X, Z, nacq[b])
            cell[b, te] = (base if b == 0 else base[..., None] * np.ones(nacq[b])) * (1 + 0.01 * rng.standard_normal(shape))
    matio.savemat(path, {"hybrid_raw": cell, "b": np.array([0.0, 150.0, 1000.0, 1500.0]), "TE": np.array([0.0, 13.0, 93.0, 143.0])})
    return cell


def test_superresHybrid_entry_point(tmp_path):
    """superresHybrid.py: master.mat in -> recon_hybrid, compartment maps, ADC map, cancer map out; the maps equal a direct
    drivers.fit_hybrid call with the same seed on the acquisition-averaged volume."""
    from mri_super_resolution_amd.scripts import superresHybrid as hyb
    os.makedirs(str(tmp_path / "pat099"))
    path = str(tmp_path / "pat099" / "master.mat")
    cell = _hybrid_master(path)
    out = str(tmp_path / "res")
    argv = ["--data", path, "--output_address", out, "--number_of_epochs", "60", "--hidden_dim", "128", "--num_layers", "2",
            "--mapping_size", "32", "--roi_start_x", "4", "--roi_end_x", "28", "--roi_start_y", "4", "--roi_end_y", "28",
            "--slice", "1", "--seed", "0"]
    s = hyb.main(argv)
    d = os.path.join(out, "pat099")
    assert s["pt_id"] == "099" or s["pt_id"] == "99"
    assert open(os.path.join(d, "ssim_scores.csv")).read() == hyb.SSIM_HEADER_HYBRID
    saved = matio.loadmat(os.path.join(d, "hybrid.mat"))
    assert saved["recon_hybrid"].shape == (48, 48, 3, 4, 4) and saved["D"].shape == (48, 48, 3) and saved["v"].shape == (48, 48, 3)
    assert saved["adc_map"].shape == (48, 48) and saved["cancer_map"].shape == (48, 48)
    assert np.array_equal(np.load(os.path.join(d, "recon_hybrid.npy")), saved["recon_hybrid"])
    raw, bvals, te = hyb.load_hybrid(path)
    assert raw.shape == (32, 32, 3, 4, 4) and list(bvals) == [0.0, 150.0, 1000.0, 1500.0] and list(te) == [0.0, 13.0, 93.0, 143.0]
    assert np.allclose(raw[..., 2, 1], np.asarray(cell[2, 1]).mean(-1), rtol=1e-6)                  # superresHybrid.py:51-54
    direct = drivers.fit_hybrid(raw, roi=(4, 28, 4, 28), slice_index=1, steps=60, seed=0, hidden_features=128,
                                hidden_layers=2, mapping_size=32, ff_scale=0.5)
    assert np.allclose(saved["v"], direct["v"], atol=1e-6) and np.allclose(saved["D"], direct["D"], atol=1e-6)
    want_adc = inr.calculate_ADC(bvals, np.squeeze(saved["recon_hybrid"][:, :, 1, :, 0]))
    assert np.allclose(saved["adc_map"], want_adc)
    # remove_small_objects: 4-connected components under 12 pixels vanish, larger ones stay whole
    m = np.zeros((12, 12), bool)
    m[1:4, 1:4] = True            # 9 pixels
    m[6:10, 5:9] = True           # 16 pixels
    m[10, 9] = True               # touches the big block only diagonally: its own 1-pixel object
    out_m = hyb.remove_small_objects(m, 12)
    assert out_m[6:10, 5:9].all() and out_m.sum() == 16
    met = json.load(open(os.path.join(d, "metrics.json")))
    assert met["recon_shape"] == [48, 48, 3, 4, 4] and met["steps"] == 60 and 0.0 <= met["voxel_fits_converged"] <= 1.0


def test_superresDWI_two_ranks_split_the_patient_list(tmp_path, golden):
    """The patient loop under torchrun (2 ranks on the one test GPU, gloo standing in for RCCL): the three patients are dealt
    over the ranks, every rank writes its own outputs, rank 0 prints the gathered summaries."""
    import socket
    import subprocess
    import sys
    rest = golden("patients_mean_b0.npz")
    files = []
    for name in ("pat41", "pat76", "pat08"):
        p = str(tmp_path / f"{name}_mean_b0.mat")
        matio.savemat(p, {"data_mean_b0": rest[name].astype(np.float32)})
        files.append(p)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "mri-super-resolution_amd", "scripts", "superresDWI.py")
    out_dir = str(tmp_path / "res")
    env = dict(os.environ, INR_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), script, "--data", *files, "--output_address", out_dir,
                          "--number_of_epochs", "40", "--seed", "0"], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, run.stderr[-3000:]
    line = [l for l in run.stdout.splitlines() if l.startswith('{"patients"')]
    assert len(line) == 1, run.stdout[-2000:]
    pats = json.loads(line[0])["patients"]
    assert [p["pt_id"] for p in pats] == ["41", "76", "08"] and sorted({p["rank"] for p in pats}) == [0, 1]
    assert [int(p["n_coords"]) for p in pats] == [25 * 25 * 24, 25 * 25 * 34, 25 * 25 * 28]
    for p in pats:
        d = os.path.join(out_dir, f"pat{p['pt_id']}")
        m = json.load(open(os.path.join(d, "metrics.json")))
        assert m["psnr_db"] == pytest.approx(p["psnr_db"], rel=1e-9) and os.path.exists(os.path.join(d, "recon.mat"))
