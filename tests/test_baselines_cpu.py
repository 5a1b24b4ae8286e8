"""CPU: host-side pieces either side of the hot path -- .mat container, CSV schemas, contrast metric, and the oracle's
restatement of the spline baseline (pinned against the scipy call skimage 0.20 makes)."""
import importlib.util
import os
import types

import numpy as np
import pytest
import scipy.io as sio
import scipy.ndimage as ndi

from oracle import inr_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    """Host-only modules of the package, loaded by path (importing the package proper needs the HIP library)."""
    spec = importlib.util.spec_from_file_location(f"_hostonly_{name}", os.path.join(ROOT, "mri-super-resolution_amd", f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


matio = _load("matio")
reports = _load("reports")


@pytest.mark.parametrize("shape,scale", [((25, 25), 2), ((25, 25), 4), ((50, 50), 2), ((7, 13), 3), ((2, 2), 2), ((1, 5), 2)])
def test_rescale_restatement_matches_scipy_zoom(shape, scale):
    img = np.random.default_rng(sum(shape) + scale).random(shape)
    want = ndi.zoom(img, scale, order=1, mode="mirror", grid_mode=True)
    got = O.rescale_linear(img, scale)
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=0, atol=1e-13)


def test_contrast_restatement_matches_reference_fixture(golden):
    g = golden("contrast.npz")
    for k in range(int(g["count"])):
        scale, focus = (int(v) for v in g[f"c{k}/scale_focus"])
        locs = g[f"c{k}/locs"]
        got = O.calculate_contrast(locs[0], locs[1], locs[2], scale, g[f"c{k}/image"], focus)
        assert np.allclose(got, g[f"c{k}/result"], rtol=1e-12)
        # the package's own function (host-side, the name master.py imports) on the same reference outputs
        from types import SimpleNamespace
        from mri_super_resolution_amd import contrast
        rec = SimpleNamespace(cancer_loc=tuple(locs[0]), contralateral_loc=tuple(locs[1]), noise=tuple(locs[2]))
        assert np.allclose(contrast.calculate_contrast(rec, scale, g[f"c{k}/image"], focus), g[f"c{k}/result"], rtol=1e-12)


def test_mat5_reader_against_scipy_writer(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.random((5, 7, 3)).astype(np.float32)
    b = rng.integers(0, 100, (4, 6)).astype(np.int16)
    cell = np.empty((4, 4), dtype=object)                       # hybrid_raw-like: [b][te] cell of volumes
    for i in range(4):
        for j in range(4):
            cell[i, j] = rng.random((3, 2, 2, i + 1))
    for comp in (False, True):
        path = str(tmp_path / f"s{int(comp)}.mat")
        sio.savemat(path, {"data": a, "idx": b, "hybrid_raw": cell, "b": np.array([0.0, 150.0, 1000.0, 1500.0])},
                    do_compression=comp)
        r = matio.loadmat(path)
        assert np.array_equal(r["data"], a) and r["data"].dtype == np.float32
        assert np.array_equal(r["idx"], b) and r["idx"].dtype == np.int16
        assert all(np.array_equal(r["hybrid_raw"][i][j], cell[i, j]) for i in range(4) for j in range(4))
        assert np.array_equal(r["b"].reshape(-1), [0.0, 150.0, 1000.0, 1500.0])


def test_mat5_writer_against_scipy_reader(tmp_path):
    rng = np.random.default_rng(1)
    vol = rng.random((6, 5, 4)).astype(np.float32)
    path = str(tmp_path / "w.mat")
    matio.savemat(path, {"recon": vol, "SR_recon": vol.astype(np.float64) * 2, "shape": np.array([6, 5, 4], np.int32),
                         "cells": [vol[:, :, 0], vol[:, :, 1]]})
    r = sio.loadmat(path)
    assert np.array_equal(r["recon"], vol) and r["recon"].dtype == np.float32
    assert np.array_equal(r["SR_recon"], vol.astype(np.float64) * 2)
    assert np.array_equal(r["shape"].reshape(-1), [6, 5, 4])
    assert np.array_equal(r["cells"][0, 1], vol[:, :, 1])
    assert np.array_equal(matio.loadmat(path)["recon"], vol)            # own round trip
    with open(str(tmp_path / "h5.mat"), "wb") as fh:
        fh.write(b"MATLAB 7.3 MAT-file" + b" " * 200)
    with pytest.raises(matio.MatFormatError):
        matio.loadmat(str(tmp_path / "h5.mat"))


def test_csv_schemas(tmp_path):
    with reports.SsimCsv(str(tmp_path / "r" / "ssim_scores.csv")) as f:
        f.row(65, 150.0, 3, 0.91, 0.88)
    text = open(str(tmp_path / "r" / "ssim_scores.csv")).read()
    assert text == "Pt_id, b-value, slice, SSIM-spline, SSIM-SR\n65, 150.0, 3, 0.91, 0.88\n"      # superresDWI.py:27,186
    c = reports.ContrastCsv(str(tmp_path / "sr2.csv"))
    c.rows(0, "07", "x", {"mean": 1, "superres": 2}, lambda im: (im, im * 2, im * 3))
    rows = reports.read_csv(str(tmp_path / "sr2.csv"))
    assert open(str(tmp_path / "sr2.csv")).readline() == "seed,patient,direction,image,metric,performance\n"  # master.py:62
    assert len(rows) == 6 and rows[4] == {"seed": "0", "patient": "07", "direction": "x", "image": "superres",
                                          "metric": "CNR", "performance": "4"}
