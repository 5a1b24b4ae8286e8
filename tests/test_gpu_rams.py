"""-m gpu: RAMS forward on the HIP kernels against the torch-CPU restatement (oracle/rams_port.py).
Parity with the TensorFlow reference itself is UNPINNED (TF absent, checkpoints stripped) -- see the oracle header."""
import random

import numpy as np
import pytest
import torch

from mri_super_resolution_amd import rams
from oracle import inr_oracle as O
from oracle import rams_port as R

pytestmark = pytest.mark.gpu


def test_layer_specs_agree_with_oracle():
    assert rams.rams_layer_specs() == R.rams_layer_specs()
    assert len(rams.rams_layer_specs()) == 71


@pytest.mark.parametrize("B,H,W", [(1, 24, 20), (3, 16, 16)])
def test_forward_matches_oracle(B, H, W):
    params = R.init_rams_params(seed=1, perturb_g=True)
    model = rams.RAMS(3, 32, 3, 9, 8, 12, params=params)
    x = (np.random.default_rng(B).random((B, H, W, 9)) * 20000).astype(np.float32)
    want = R.rams_forward(params, x)
    got = model(x).cpu().numpy()
    assert got.shape == (B, 3 * H, 3 * W, 1)
    # de-normalised output carries the +7433 offset: compare the normalised residual as well
    assert O.rel_l2(got, want) < 1e-5
    assert O.rel_l2((got - R.MEAN) / R.STD, (want - R.MEAN) / R.STD) < 5e-5
    pt = rams.predict_tensor(model, x).cpu().numpy()
    ref = R.predict_tensor(params, x)
    # outputs are ~1e4 with ~1e-6 relative error: a value within ~1e-2 of a half-integer may round the other way
    assert (pt != ref).mean() < 1e-2
    assert np.abs(pt - ref).max() <= 1.0
    assert pt.min() >= 0 and pt.max() <= 65536


def test_smaller_network_and_default_init():
    model = rams.RAMS(3, 32, 3, 9, 8, 2, seed=3)         # N = 2 RFABs, Keras-style default init
    x = (np.random.default_rng(0).random((2, 12, 14, 9)) * 65535).astype(np.float32)
    want = R.rams_forward(model.params, x, N=2)
    got = model(x).cpu().numpy()
    assert O.rel_l2((got - R.MEAN) / R.STD, (want - R.MEAN) / R.STD) < 5e-5


def test_predict_case_protocol():
    params = R.init_rams_params(seed=2)
    model = rams.RAMS(params=params)
    stack = (np.random.default_rng(5).random((16, 16, 12)) * 200).astype(np.float32)
    mean, subsets = rams.predict_case(model, stack, sample_size=4, rng=random.Random(0))
    assert tuple(mean.shape) == (48, 48) and len(subsets) == 4 and all(len(s) == 9 for s in subsets)
    lor = stack[None].astype("uint16") * 256
    want = np.mean([R.predict_tensor(params, lor[:, :, :, s].astype(np.float32))[0, :, :, 0] for s in subsets], axis=0)
    assert np.abs(mean.cpu().numpy() - want).max() <= 0.5


def test_rejects_unsupported_configs():
    with pytest.raises(Exception):
        rams.RAMS(3, 64, 3, 9, 8, 2).pack()
    with pytest.raises(ValueError):
        rams.RAMS(seed=0)(np.zeros((1, 8, 8, 5), np.float32))


def test_shift_tolerant_losses_match_oracle():
    """cL1 / cPSNR of utils/loss.py (7x7 shifts, brightness bias, masks) against the float64 restatement."""
    rng = np.random.default_rng(7)
    B, size = 3, 60
    y_true = (rng.random((B, size, size)) * 40000 + 2000).astype(np.float32)
    y_pred = np.roll(y_true, (1, -2), axis=(1, 2)) * 0.97 + 150 + rng.standard_normal((B, size, size)).astype(np.float32) * 30
    mask = (rng.random((B, size, size)) > 0.15).astype(np.float32)
    want_l1, want_ps = R.shift_losses(y_true, y_pred, mask, size)
    got_l1 = rams.l1_loss(y_true, y_pred.astype(np.float32), mask, HR_SIZE=size).cpu().numpy()
    got_ps = rams.psnr(y_true[..., None], y_pred.astype(np.float32)[..., None], mask[..., None], size_image=size).item()
    assert np.allclose(got_l1, want_l1, rtol=1e-9)
    assert got_ps == pytest.approx(want_ps.mean(), rel=1e-10)
    # the planted shift (+1, -2 relative to the centre (3,3)) is the best one: loss far below the unshifted one
    unshifted = np.abs((y_true - y_pred)[:, 3:-3, 3:-3]).mean()
    assert (got_l1 < 0.5 * unshifted).all()
