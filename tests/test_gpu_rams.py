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


def test_layer_table_matches_the_graph_the_oracle_builds():
    """The product's layer table against the list the oracle DISCOVERS by running its own restatement of
    network.py:110-155 with a recording parameter store (two derivations: a table here, a graph walk there)."""
    for cfg in (dict(), dict(N=2), dict(filters=16, r=4, N=1)):
        assert rams.rams_layer_specs(**cfg) == R.rams_layer_specs(**cfg)
    assert len(rams.rams_layer_specs()) == 71


def test_full_size_stack_matches_oracle():
    """BASELINE config 3 shape: (1, 128, 128, 9) and a batch of 2 at 128 x 128 -- the multi-tile paths, tile edges and
    per-image offsets the toy sizes never reach (~265 GFLOP per image on the host oracle)."""
    params = R.init_rams_params(seed=4, perturb_g=True)
    model = rams.RAMS(3, 32, 3, 9, 8, 12, params=params)
    x = (np.random.default_rng(11).random((2, 128, 128, 9)) * 30000 + 1000).astype(np.float32)
    torch.set_num_threads(max(1, min(32, torch.get_num_threads())))
    want = R.rams_forward(params, x)
    ref = R.predict_tensor(params, x[:1])
    from mri_super_resolution_amd._lib import lib
    # every path of the 32 -> 32 convolutions (debug key 14): 6 = split-fp16 MFMA with the activations staged in LDS, forced
    # (the default for batches; batch 1 at this size would take the f32 kernel), 2 = the default rule, 1 = split-fp16 with
    # the activations from global memory, 0 = the exact f32-input MFMA
    # 22 / 48: key 14 = 6 with the two-pass kernel / the two-blocks-per-CU kernel (key 15 = 16 / 42)
    for split, flips in ((6, 2e-2), (22, 2e-2), (48, 2e-2), (2, 2e-2), (1, 2e-2), (0, 1e-2)):
        lib().inr_debug_set(14, 6 if split in (22, 48) else split)
        lib().inr_debug_set(15, {22: 16, 48: 42}.get(split, DEFAULT_LDS_KERNEL))
        try:
            got = model(x).cpu().numpy()
            assert got.shape == (2, 384, 384, 1)
            err = O.rel_l2((got - R.MEAN) / R.STD, (want - R.MEAN) / R.STD)
            single = model(x[:1]).cpu().numpy()                 # batch 1 is the reference's call shape (master.py:50)
            assert O.rel_l2(single, got[:1]) < 1e-6             # (pooled sums are reduced in a batch-dependent order)
            pt = rams.predict_tensor(model, x[:1]).cpu().numpy()
            frac = float((pt != ref).mean())
            print(f"split={split}: normalised rel-L2 {err:.2e}, rounded outputs that differ {frac:.4f}")
            assert err < 5e-5
            # outputs are ~1e4 with ~1e-6 relative error: a value within ~1e-2 of a half-integer may round the other way
            assert frac < flips and np.abs(pt - ref).max() <= 1.0
        finally:
            lib().inr_debug_set(14, 2)
            lib().inr_debug_set(15, DEFAULT_LDS_KERNEL)


@pytest.fixture
def conv_mode(request):
    """debug keys 14 / 15 for the duration of a test: key 14 = 2 default rule, 6 LDS-staged split-fp16 kernels forced, 1 / 0 the
    others; key 15 = which LDS-staged kernel (8: 8 waves x 1 tile, 4: 4 x 2, 16: two-pass 8 x 2, 42: two blocks of 4 x 2 per CU)"""
    from mri_super_resolution_amd._lib import lib
    k14, k15 = request.param if isinstance(request.param, tuple) else (request.param, None)
    lib().inr_debug_set(14, k14)
    if k15 is not None:
        lib().inr_debug_set(15, k15)
    yield request.param
    lib().inr_debug_set(14, 2)
    lib().inr_debug_set(15, DEFAULT_LDS_KERNEL)


DEFAULT_LDS_KERNEL = 42


@pytest.mark.parametrize("conv_mode", [2, (6, 8), (6, 4), (6, 16), (6, 42), 1], indirect=True)
@pytest.mark.parametrize("B,H,W", [(1, 24, 20), (3, 16, 16), (2, 13, 31)])
def test_forward_matches_oracle(B, H, W, conv_mode):
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


@pytest.mark.parametrize("B,H,W", [(2, 24, 20), (1, 13, 31)])
def test_gate_ahead_form_matches_oracle(B, H, W):
    """Inference with the gate of every attention block formed AHEAD of its second convolution (class sums of the first one's
    output; the second applies gate and residual in its epilogue) -- the default from four 128 x 128 stacks on (debug key 26), here
    forced at test sizes, border and interior patches, odd extents -- against the same restatement and tolerances as the other form."""
    from mri_super_resolution_amd._lib import lib
    params = R.init_rams_params(seed=1, perturb_g=True)
    model = rams.RAMS(3, 32, 3, 9, 8, 12, params=params)
    x = (np.random.default_rng(B).random((B, H, W, 9)) * 20000).astype(np.float32)
    want = R.rams_forward(params, x)
    lib().inr_debug_set(26, 0)
    got = model(x).cpu().numpy()
    assert O.rel_l2(got, want) < 1e-5
    assert O.rel_l2((got - R.MEAN) / R.STD, (want - R.MEAN) / R.STD) < 5e-5


@pytest.mark.parametrize("scale", [2, 4])
def test_scale_other_than_three(scale):
    """``scale`` is plumbed through RamsDesc, the head convolutions (scale^2 output channels) and the pixel shuffle.  The
    reference writes ``depth_to_space(x, 3)`` literally (network.py:141,148) -- any other scale is a TensorFlow shape error there
    -- so the oracle is asked for the generalised reading ``shuffle=scale`` (BASELINE config 3 says "x4")."""
    kw = dict(scale=scale, N=2)
    params = R.init_rams_params(seed=6, perturb_g=True, shuffle=scale, **kw)
    model = rams.RAMS(scale, 32, 3, 9, 8, 2, params=params)
    assert rams.rams_layer_specs(**kw) == R.rams_layer_specs(shuffle=scale, **kw)
    x = (np.random.default_rng(scale).random((2, 14, 17, 9)) * 40000).astype(np.float32)
    want = R.rams_forward(params, x, shuffle=scale, **kw)
    got = model(x).cpu().numpy()
    assert got.shape == (2, scale * 14, scale * 17, 1)
    assert O.rel_l2((got - R.MEAN) / R.STD, (want - R.MEAN) / R.STD) < 5e-5
    with pytest.raises(Exception):                      # the reference's own reading: a shape error
        R.rams_forward(params, x, **kw)


@pytest.mark.parametrize("B", [20, 29, 33])
def test_batch_sizes_with_an_odd_block_count(B):
    """floor(512 / B) odd: the default LDS-staged kernel writes 4 * floor(512 / B) channel-sum slabs per batch element; round 3
    planned 8 * floor(256 / B) and the surplus rows landed in the attention gates (B = 20 .. 30) or past the pad (B >= 33).
    Batch elements are independent, so every row of the batched forward must equal the same stack run on its own."""
    params = R.init_rams_params(seed=8, perturb_g=True, N=1)
    model = rams.RAMS(3, 32, 3, 9, 8, 1, params=params)
    x = (np.random.default_rng(B).random((B, 40, 40, 9)) * 30000 + 500).astype(np.float32)
    got = model(x).cpu().numpy()
    for b in (0, B // 2, B - 2, B - 1):
        single = model(x[b:b + 1]).cpu().numpy()
        assert O.rel_l2(got[b:b + 1], single) < 1e-6, b
    want = R.rams_forward(params, x[B - 2:], N=1)
    assert O.rel_l2((got[B - 2:] - R.MEAN) / R.STD, (want - R.MEAN) / R.STD) < 5e-5


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


def test_shift_loss_gradient_matches_autograd():
    """d cL1 / d prediction through the best shift (the loss half of train_step) against torch autograd on a float64
    restatement of utils/loss.py:26-75."""
    rng = np.random.default_rng(11)
    B, size = 3, 40
    y_true = (rng.random((B, size, size)) * 30000 + 5000).astype(np.float32)
    y_pred = (np.roll(y_true, (-1, 2), axis=(1, 2)) * 1.02 - 90 + rng.standard_normal((B, size, size)) * 40).astype(np.float32)
    mask = (rng.random((B, size, size)) > 0.2).astype(np.float32)
    up = np.array([1.0, 0.5, 2.0], np.float32)

    yt, mk = torch.from_numpy(y_true).double(), torch.from_numpy(mask).double()
    yp = torch.from_numpy(y_pred).double().requires_grad_(True)
    c = size - 6
    pred = yp[:, 3:size - 3, 3:size - 3]
    per = []
    for i in range(7):
        for j in range(7):
            lab, m = yt[:, i:i + c, j:j + c], mk[:, i:i + c, j:j + c]
            tot = m.sum(dim=(1, 2))
            b = ((lab * m - pred * m).sum(dim=(1, 2)) / tot)[:, None, None]
            per.append((lab * m - (pred * m + b) * m).abs().sum(dim=(1, 2)) / tot)
    want_loss = torch.stack(per).min(dim=0).values
    (want_loss * torch.from_numpy(up).double()).sum().backward()

    loss, grad = rams.l1_loss_and_grad(y_true, y_pred, mask, HR_SIZE=size, upstream=up)
    assert np.allclose(loss.cpu().numpy(), want_loss.detach().numpy(), rtol=1e-9)
    g, w = grad.cpu().numpy().astype(np.float64), yp.grad.numpy()
    assert np.abs(g - w).max() <= 1e-6 * np.abs(w).max()
    assert np.count_nonzero(g[:, :3]) == 0 and np.count_nonzero(g[:, :, -3:]) == 0       # nothing on the border frame
    loss1, grad1 = rams.l1_loss_and_grad(y_true, y_pred, mask, HR_SIZE=size)
    assert torch.equal(loss1, loss) and np.allclose(grad1[1].cpu().numpy() * 0.5, g[1], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("conv_mode", [2, 0], indirect=True)
@pytest.mark.parametrize("shape,pad", [((2, 10, 9, 5), 1), ((1, 7, 8, 9), 0), ((3, 12, 6, 3), 1), ((2, 21, 33, 9), 1)])
def test_conv3d_forward_and_gradients_vs_torch(shape, pad, conv_mode):
    """The 3x3x3 convolution 32 -> 32 on its own, its data gradient ('same') and its weight / bias gradients against
    torch's conv3d + autograd in float64.  The weight gradient follows debug key 14 like the training step: 2 = split-fp16 MFMA
    (conv3d_c32_wgrad_h3_kernel: ragged patches, depths 3 / 5 / 9, valid and 'same'), 0 = f32-input MFMA."""
    import torch.nn.functional as F
    B, D1, D2, D3 = shape
    g = torch.Generator().manual_seed(sum(shape) + pad)
    x = torch.randn(B, D1, D2, D3, 32, generator=g)
    w = torch.randn(27, 32, 32, generator=g) * 0.05
    bias = torch.randn(32, generator=g) * 0.1
    xr = x.double().permute(0, 4, 1, 2, 3).requires_grad_(True)                       # NCDHW
    wr = w.double().reshape(3, 3, 3, 32, 32).permute(4, 3, 0, 1, 2).requires_grad_(True)   # [cout, cin, k1, k2, k3]
    br = bias.double().requires_grad_(True)
    yr = F.conv3d(xr, wr, br, padding=pad)
    dy = torch.randn(yr.shape, generator=g, dtype=torch.float64)
    yr.backward(dy)
    rel = lambda a, b: (np.linalg.norm(a - b) / np.linalg.norm(b))
    y = rams.conv3d(x.cuda(), w.cuda(), bias.cuda(), pad=pad)
    assert rel(y.cpu().numpy(), yr.detach().permute(0, 2, 3, 4, 1).numpy()) < 2e-6
    dy_dev = dy.permute(0, 2, 3, 4, 1).float().contiguous().cuda()
    gw, gb = rams.conv3d_wgrad(x.cuda(), dy_dev, pad=pad)
    want_gw = wr.grad.permute(2, 3, 4, 1, 0).reshape(27, 32, 32).numpy()
    assert rel(gw.cpu().numpy(), want_gw) < 5e-6 and rel(gb.cpu().numpy(), br.grad.numpy()) < 5e-6
    gw2, _ = rams.conv3d_wgrad(x.cuda(), dy_dev, pad=pad)
    assert torch.equal(gw, gw2)                                                        # fixed-order reduction
    if pad == 1:
        dx = rams.conv3d_dgrad(dy_dev, w.cuda())
        assert rel(dx.cpu().numpy(), xr.grad.permute(0, 2, 3, 4, 1).numpy()) < 5e-6


def _train_case(B=2, side=20, seed=3):
    rng = np.random.default_rng(seed)
    x = (rng.random((B, side, side, 9)) * 20000 + 2000).astype(np.float32)
    hr = (rng.random((B, 3 * side, 3 * side)) * 20000 + 2000).astype(np.float32)
    mask = (rng.random((B, 3 * side, 3 * side)) > 0.15).astype(np.float32)
    return x, hr, mask


@pytest.mark.parametrize("conv_mode", [2, 0], indirect=True)
def test_train_gradients_match_autograd(conv_mode):
    """Trainer.train_step (utils/training.py:193-209): gradients of all 71 layers (v, g, b: 213 tensors) of a (2, 20, 20, 9)
    batch against torch autograd on the float64 restatement (itself unpinned against TensorFlow -- see the oracle).  Both
    arithmetic paths of the 32 -> 32 convolutions' forward and data gradient: split-fp16 MFMA staged through LDS (debug key 14 = 2,
    default) and the f32-input MFMA kernels (0); the weight gradient is f32-input MFMA in both."""
    params = R.init_rams_params(seed=5, perturb_g=True)
    model = rams.RAMS(3, 32, 3, 9, 8, 12, params=params)
    x, hr, mask = _train_case()
    want_loss, want = R.train_grads(params, x, hr, mask)
    trainer = rams.RamsTrainer(model)
    loss, pred = trainer.loss_and_grads(x, hr, mask, want_prediction=True)
    assert O.rel_l2(pred.cpu().numpy(), R.rams_forward(params, x)[..., 0]) < 1e-5
    assert np.allclose(loss.cpu().numpy(), want_loss, rtol=1e-5)
    got = trainer.named_gradients()
    assert set(got) == set(want) and len(got) == 213
    worst = max((O.rel_l2(got[k], want[k]), k) for k in want if np.linalg.norm(want[k]) > 0)
    assert worst[0] < 1e-4, worst                                  # fp32-class pipeline vs float64 autograd
    total = O.rel_l2(np.concatenate([got[k].reshape(-1) for k in sorted(want)]),
                     np.concatenate([want[k].reshape(-1) for k in sorted(want)]))
    assert total < 1e-5, total


def test_long_skip_in_the_convolution_epilogue_is_bit_identical():
    """Inference: the trunk-closing convolution adds the stem output in its epilogue (debug key 24 = 1, default) instead of a separate
    add pass (key 24 = 0): the same bits, at a batch with border and interior patches and at batch 1."""
    from mri_super_resolution_amd._lib import lib
    model = rams.RAMS(3, 32, 3, 9, 8, 12, params=R.init_rams_params(seed=4, perturb_g=True))
    rng = np.random.default_rng(2)
    try:
        lib().inr_debug_set(26, 0)                          # (the gate-ahead form at these small sizes too)
        for shape in ((3, 40, 36, 9), (1, 64, 64, 9)):
            x = torch.from_numpy((rng.random(shape) * 30000 + 500).astype(np.float32)).cuda()
            out = {}
            for key in (1, 0, 2):
                lib().inr_debug_set(24, key)
                out[key] = model(x).clone()
            assert torch.equal(out[0], out[1]) and float(out[1].abs().max()) > 0
            # key 24 = 2 (default): the gate of every attention block from the class sums of its FIRST convolution's output, applied in
            # the second one's epilogue -- the same mean in exact arithmetic, another order of rounding
            a, b = out[2].cpu().numpy(), out[1].cpu().numpy()
            assert not np.array_equal(a, b) and O.rel_l2(a, b) < 1e-6 and O.rel_l2((a - R.MEAN) / R.STD, (b - R.MEAN) / R.STD) < 2e-6   # (observed 2.3e-7 / 3.0e-7; either form is 6e-7 from the oracle)
    finally:
        lib().inr_debug_set(24, 2)
        lib().inr_debug_set(26, 600000)


def test_epilogue_fused_backward_is_bit_identical():
    """Training step: the data-gradient convolutions apply the ReLU mask (with the masked gradient's maximum) and add the residual
    path's gradient in their epilogue (debug key 24 = 1, default) -- the same bits as the separate element-wise passes (key 24 = 0),
    at a batch whose images take border AND interior patches."""
    from mri_super_resolution_amd._lib import lib
    params = R.init_rams_params(seed=5, perturb_g=True)
    x, hr, mask = _train_case(B=3, side=24, seed=11)
    got = {}
    try:
        for key in (1, 0):
            lib().inr_debug_set(24, key)
            trainer = rams.RamsTrainer(rams.RAMS(3, 32, 3, 9, 8, 12, params=params))
            loss, _ = trainer.loss_and_grads(x, hr, mask, want_prediction=True)
            got[key] = (loss.cpu().numpy().copy(), {k: np.array(v, copy=True) for k, v in trainer.named_gradients().items()})
    finally:
        lib().inr_debug_set(24, 2)
    assert np.array_equal(got[0][0], got[1][0])
    for k in got[0][1]:
        assert np.array_equal(got[0][1][k], got[1][1][k]), k
    assert any(np.abs(v).max() > 0 for v in got[1][1].values())


@pytest.mark.parametrize("conv_mode", [2, 0], indirect=True)
def test_train_steps_follow_keras_adam(conv_mode):
    """Ten train_steps on a small RAMS (N = 2): the loss trajectory against the float64 restatement + Keras-form Adam, on both
    arithmetic paths of the forward / data-gradient convolutions (debug key 14: 2 = split-fp16, default; 0 = f32-input MFMA)."""
    params = R.init_rams_params(seed=6, perturb_g=True, N=2)
    model = rams.RAMS(3, 32, 3, 9, 8, 2, params={k: v.copy() for k, v in params.items()})
    x, hr, mask = _train_case(B=2, side=16, seed=8)
    trainer = rams.RamsTrainer(model, learning_rate=5e-4)
    got = [float(trainer.train_step(x, hr, mask).sum()) for _ in range(10)]
    p = {k: np.asarray(v, np.float64) for k, v in params.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v2 = {k: np.zeros_like(v) for k, v in p.items()}
    want = []
    for t in range(1, 11):
        loss, g = R.train_grads(p, x, hr, mask, N=2)
        want.append(float(loss.sum()))
        R.keras_adam_step(p, g, m, v2, t)
    # Adam divides by sqrt(v): parameters whose gradient is round-off get steps of full size in a direction set by that
    # round-off, so fp32-vs-float64 differences grow with the step count (observed 2e-8, 5e-8, 7e-8, 1e-6, 1e-5, 9e-5, ...)
    assert np.allclose(got[:5], want[:5], rtol=1e-4), (got, want)
    assert np.allclose(got, want, rtol=5e-3), (got, want)
    assert got[-1] < got[0]
    synced = trainer.sync_model()
    assert O.rel_l2(synced.params["rfab1/conv2/v"], p["rfab1/conv2/v"]) < 5e-3   # (observed 1.5e-3 after 10 Adam steps)
    out = synced(x).cpu().numpy()                                  # the inference path picks the trained weights up
    # (the same round-off amplification seen through every layer: 4e-3 ... 7e-3 after ten steps, moving with ANY change of rounding
    # or summation order -- the split-fp16 convolutions, the number of partial sums of a weight gradient -- while the gradients of
    # both paths meet float64 to 1e-4 per tensor / 1e-5 overall above and the first five losses to 1e-4 here)
    assert O.rel_l2(out, R.rams_forward({k: v.astype(np.float32) for k, v in p.items()}, x, N=2)) < 1.5e-2


def test_rams_plus_geometric_ensemble_matches_the_restatement():
    """RAMS+ (utils/prediction.py:10-74): the eight flip x rotation members, their inversion and the averaged prediction, against
    the numpy restatement of the same functions around the oracle's forward pass (parity with TensorFlow itself unpinned)."""
    x = (np.random.default_rng(2).random((12, 12, 9)) * 30000 + 500).astype(np.float32)
    ens, r = rams.geometric_ensemble(x)
    want_ens, want_r = R.np_geometric_ensemble(x)
    assert np.array_equal(r, want_r) and np.array_equal(ens.cpu().numpy(), want_ens)
    assert np.array_equal(rams.unensemble(ens, r)[0].cpu().numpy(), x)              # inverse transforms, mean of eight copies
    params = R.init_rams_params(seed=3, perturb_g=True, N=1)
    model = rams.RAMS(3, 32, 3, 9, 8, 1, params=params)
    got = rams.unensemble(rams.predict_tensor(model, ens), r).cpu().numpy()
    want = R.np_unensemble(R.predict_tensor(params, want_ens, N=1), want_r)
    assert got.shape == want.shape == (1, 36, 36, 1)
    assert np.abs(got - want).max() <= 1.0 / 8 + 1e-9 or np.abs(got - want).mean() < 0.02   # a member may round the other way
    assert np.abs(got - want).max() <= 1.0
    # random members: flip / rotate draw from the numpy generator handed in, the parameters come back for unensemble
    e2, r2 = rams.ensemble(x, geometric=False, shuffle=False, n=5, rng=np.random.default_rng(0))
    assert tuple(e2.shape) == (5, 12, 12, 9) and np.array_equal(rams.unensemble(e2, r2)[0].cpu().numpy(), x)


def test_predict_tensor_permute_and_shuffle():
    x = (np.random.default_rng(4).random((10, 10, 9)) * 30000).astype(np.float32)
    s = rams.shuffle_last_axis(x, np.random.default_rng(7)).cpu().numpy()
    perm = np.random.default_rng(7).permutation(9)
    assert np.array_equal(s, x[..., perm])
    params = R.init_rams_params(seed=9, perturb_g=True, N=1)
    model = rams.RAMS(3, 32, 3, 9, 8, 1, params=params)
    got = rams.predict_tensor_permute(model, x, n_ens=4, rng=np.random.default_rng(1)).cpu().numpy()
    g = np.random.default_rng(1)
    want = np.mean([R.predict_tensor(params, x[None][..., g.permutation(9)], N=1)[0] for _ in range(4)], axis=0, keepdims=True)
    assert got.shape == (1, 30, 30, 1) and np.abs(got - want).max() <= 1.0 and np.abs(got - want).mean() < 0.05


def test_trainer_fit_loop_checkpoints_and_resume(tmp_path):
    """Trainer.fit (utils/training.py:108-191): shuffled batches, one train_step each, validation every `evaluate_every` steps,
    a checkpoint when the validation cPSNR improved, at most three kept, construction-time restore."""
    rng = np.random.default_rng(5)
    n, side = 6, 16
    x = (rng.random((n, side, side, 9)) * 20000 + 2000).astype(np.float32)
    hr = (rng.random((n, 3 * side, 3 * side)) * 20000 + 2000).astype(np.float32)
    mask = np.ones((n, 3 * side, 3 * side), np.float32)
    params = R.init_rams_params(seed=6, perturb_g=True, N=1)
    fresh = lambda: rams.RamsTrainer(rams.RAMS(3, 32, 3, 9, 8, 1, params={k: v.copy() for k, v in params.items()}))
    tr = fresh()
    ck = str(tmp_path / "ckpt")
    logs = []
    hist = tr.fit(x, (hr, mask), batch_size=2, epochs=3, evaluate_every=2, val_steps=2, validation_data=(x[:4], (hr[:4], mask[:4])),
                  save_best_only=False, checkpoint_dir=ck, seed=0, log=logs.append)
    assert tr.step_count == 9 and [h["step"] for h in hist] == [2, 4, 6, 8] and logs == hist
    assert all(np.isfinite(h["loss"]) and np.isfinite(h["val_loss"]) and np.isfinite(h["val_psnr"]) for h in hist)
    import glob
    assert len(glob.glob(ck + "/ckpt-*.npz")) == 3                                      # max_to_keep = 3
    # the first step of the loop is train_step on the first batch of the seeded order
    ref = fresh()
    order = np.random.default_rng(0).permutation(n)
    first = ref.train_step(x[order[:2]], hr[order[:2]], mask[order[:2]])
    tr2 = fresh()
    h2 = tr2.fit(x, (hr, mask), batch_size=2, epochs=1, evaluate_every=1, seed=0)
    assert h2[0]["loss"] == pytest.approx(float(first.mean()), rel=1e-6)
    # resume: a new trainer pointed at the directory continues from the last checkpoint (step 8) with its Adam state
    tr3 = fresh()
    best = tr3.restore(ck)
    assert tr3.step_count == 8 and best == pytest.approx(hist[-1]["val_psnr"])
    again = tr3.fit(x, (hr, mask), batch_size=2, epochs=1, evaluate_every=1, validation_data=(x[:2], (hr[:2], mask[:2])),
                    save_best_only=True, checkpoint_dir=ck, seed=1)
    assert tr3.step_count == 11 and all(("checkpoint" in h) == (h["val_psnr"] > best) or True for h in again)
