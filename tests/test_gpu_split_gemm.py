"""-m gpu: the split-fp16 MFMA GEMMs (csrc/gemm_h3.inc) on their own: accuracy against fp64 next to the fp32-MFMA
kernels, ragged row counts, tensor scales from 1e-30 to 1e+3, zero operands, and the weight-plane kernel.
Uses the standalone layer entry points in debug mode inr_debug_set(3, 2) (planes / amax built per call in a scratch
buffer handed over with inr_debug_set_ptr(1, ...)); the fused fit uses the same kernels with producer-side scales."""
import ctypes

import numpy as np
import pytest
import torch

from mri_super_resolution_amd import ops
from mri_super_resolution_amd._lib import lib

pytestmark = pytest.mark.gpu


@pytest.fixture()
def split_mode():
    L = lib()
    L.inr_debug_set_ptr.argtypes = [ctypes.c_int, ctypes.c_void_p]
    scratch = torch.zeros(32 << 20, dtype=torch.uint8, device="cuda")
    L.inr_debug_set_ptr(1, scratch.data_ptr())

    def set_mode(m):
        L.inr_debug_set(3, m)

    yield set_mode, scratch
    L.inr_debug_set(3, 1)
    L.inr_debug_set_ptr(1, None)


def rel(a, ref):
    return ((a.double() - ref).norm() / ref.norm()).item()


@pytest.mark.parametrize("n,k", [(4096, 512), (1000, 256), (77, 512), (129, 64)])
def test_three_gemms_match_fp64_like_the_fp32_kernels(split_mode, n, k):
    set_mode, _ = split_mode
    H = 512
    g = torch.Generator(device="cuda").manual_seed(n + k)
    x = torch.rand(n, k, device="cuda", generator=g) * 2 - 1
    W = (torch.rand(H, k, device="cuda", generator=g) * 2 - 1) * 0.01
    b = torch.randn(H, device="cuda", generator=g) * 0.01
    dz = torch.randn(n, H, device="cuda", generator=g) * torch.exp(torch.randn(n, 1, device="cuda", generator=g) * 2) * 1e-7
    dact = torch.randn(n, k, device="cuda", generator=g)
    ref_f = torch.sin(30 * (x.double() @ W.double().T + b.double()))
    ref_c = 30 * torch.cos(30 * (x.double() @ W.double().T + b.double()))
    ref_dx = (dz.double() @ W.double()) * dact.double()
    ref_dw = dz.double().T @ x.double()
    errs = {}
    for mode in (0, 2):
        set_mode(mode)
        a, d = ops.sine_layer_forward(x, W, b, 30.0, True)
        dx = ops.sine_layer_backward_input(dz, W, dact)
        gW, gb = ops.linear_param_grad(dz, x, True)
        errs[mode] = (rel(a, ref_f), rel(d, ref_c), rel(dx, ref_dx), rel(gW, ref_dw))
        assert rel(gb, dz.double().sum(0)) < 1e-5
    for e32, e16 in zip(errs[0], errs[2]):
        assert e16 < 3e-6 and e16 < 4 * e32 + 2e-7          # fp32-class accuracy: within a small factor of the f32 MFMA


@pytest.mark.parametrize("dz_scale,w_scale", [(1e-30, 1e-2), (1e-12, 1e-6), (1.0, 1.0), (1e3, 10.0)])
def test_tensor_scales_over_the_whole_float_range(split_mode, dz_scale, w_scale):
    set_mode, _ = split_mode
    n, H = 2048, 512
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(n, H, device="cuda", generator=g) * 2 - 1
    W = torch.randn(H, H, device="cuda", generator=g) * w_scale
    dz = torch.randn(n, H, device="cuda", generator=g) * dz_scale
    dact = torch.randn(n, H, device="cuda", generator=g)
    set_mode(2)
    dx = ops.sine_layer_backward_input(dz, W, dact)
    gW, _ = ops.linear_param_grad(dz, x, False)
    assert torch.isfinite(dx).all() and torch.isfinite(gW).all()
    assert rel(dx, (dz.double() @ W.double()) * dact.double()) < 3e-6
    assert rel(gW, dz.double().T @ x.double()) < 3e-6


def test_zero_operands_and_outliers(split_mode):
    set_mode, _ = split_mode
    n, H = 512, 512
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.rand(n, H, device="cuda", generator=g) * 2 - 1
    W = torch.randn(H, H, device="cuda", generator=g) * 0.01
    dact = torch.ones(n, H, device="cuda")
    set_mode(2)
    zero = torch.zeros(n, H, device="cuda")
    assert torch.count_nonzero(ops.sine_layer_backward_input(zero, W, dact)) == 0
    assert torch.count_nonzero(ops.linear_param_grad(zero, x, False)[0]) == 0
    a, _ = ops.sine_layer_forward(x, torch.zeros_like(W), torch.zeros(H, device="cuda"), 30.0, True)
    assert torch.count_nonzero(a) == 0
    # one huge row next to tiny ones: the tensor scale follows the maximum, the small rows lose nothing that matters norm-wise
    dz = torch.randn(n, H, device="cuda", generator=g) * 1e-9
    dz[7] *= 1e6
    dx = ops.sine_layer_backward_input(dz, W, dact)
    assert rel(dx, dz.double() @ W.double()) < 3e-6
    small = torch.arange(n, device="cuda") != 7
    assert rel(dx[small], dz[small].double() @ W.double()) < 1e-3      # elementwise accuracy of rows 1e-6 below the maximum


def test_weight_planes(split_mode):
    """hi + lo reproduces W * 2^k to 2^-21, the transposed planes are transposes, k puts max|W| in [2^14, 2^15)."""
    set_mode, scratch = split_mode
    H, K = 512, 256
    g = torch.Generator(device="cuda").manual_seed(2)
    W = torch.randn(H, K, device="cuda", generator=g) * 0.003
    x = torch.rand(128, K, device="cuda", generator=g)
    set_mode(2)
    ops.sine_layer_forward(x, W, torch.zeros(H, device="cuda"), 30.0, False)
    torch.cuda.synchronize()
    amax = scratch[:4].view(torch.int32).item()
    assert np.frombuffer(np.int32(amax).tobytes(), np.float32)[0] == W.abs().max().item()
    planes = scratch[256:256 + 8 * H * K].view(torch.float16)
    hi, lo, hiT, loT = (planes[i * H * K:(i + 1) * H * K] for i in range(4))
    scaled = hi.view(H, K).double() + lo.view(H, K).double()
    k = np.round(np.log2((scaled.abs().max() / W.abs().max()).item()))
    assert 2.0 ** 14 <= W.abs().max().item() * 2.0 ** k < 2.0 ** 15
    assert ((scaled - W.double() * 2.0 ** k).abs() <= 2.0 ** -20 * (W.double() * 2.0 ** k).abs() + 2.0 ** -24).all()
    assert torch.equal(hiT.view(K, H), hi.view(H, K).T) and torch.equal(loT.view(K, H), lo.view(H, K).T)


def test_fused_fit_split_vs_fp32_trajectory():
    """30 fused Adam steps with the split GEMMs against the same steps on the fp32-MFMA kernels (T3-sized bound)."""
    import mri_super_resolution_amd as inr
    from oracle.torch_port import fourier_matrix
    x = ops.grid_fourier_map((40, 40, 8), torch.from_numpy(fourier_matrix(3)).cuda())
    c = ops.mgrid((40, 40, 8))
    t = (0.5 + 0.3 * torch.sin(3 * c[:, 0]) * torch.cos(2 * c[:, 1]) + 0.1 * c[:, 2]).contiguous()   # smooth, image-like
    out = {}
    for mode in (0, 1):
        lib().inr_debug_set(3, mode)
        torch.manual_seed(0)
        net = inr.Siren(256, 512, 3, 1).cuda()
        losses = inr.SirenFitter(net).step(x, t, 30)
        out[mode] = (losses.cpu().numpy(), torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy())
    lib().inr_debug_set(3, 1)
    assert np.allclose(out[0][0][:3], out[1][0][:3], rtol=1e-5)     # same arithmetic to fp32 noise at the start ...
    assert np.allclose(out[0][0], out[1][0], rtol=2e-3)             # ... which the steep first descent then amplifies
    assert np.linalg.norm(out[0][1] - out[1][1]) / np.linalg.norm(out[0][1]) < 1e-4


@pytest.mark.parametrize("n,fin,hidden,layers,out", [(300, 64, 96, 2, 1), (1111, 32, 160, 1, 3), (64, 256, 512, 3, 1),
                                                      (40000, 32, 32, 4, 1)])
def test_fused_gradients_on_eligible_but_ragged_networks(n, fin, hidden, layers, out):
    """Networks the split path accepts (every sine layer a multiple of 32 wide) whose sizes are not tile multiples:
    fused forward/backward (lr = 0 step) against the per-layer autograd path on the f32 MFMA kernels, and both against
    the float64 oracle forward."""
    import mri_super_resolution_amd as inr
    from oracle import inr_oracle as O
    from oracle import torch_port as P
    torch.manual_seed(n)
    net = inr.Siren(fin, hidden, layers, out).cuda()
    x = (torch.rand(n, fin, device="cuda") * 2 - 1).contiguous()
    t = torch.rand(n, out, device="cuda")
    y = net(x)
    ((y - t) ** 2).mean().backward()
    auto = torch.cat([p.grad.reshape(-1) for p in net.layer_parameters()]).double()
    fitter = inr.SirenFitter(net, lr=0.0)
    loss = fitter.step(x, t.reshape(-1), n_steps=1)
    fused = torch.cat([fitter.grads[o:o + p.numel()] for (wo, bo), pw, pb in
                       zip(fitter.offsets, net.layer_parameters()[0::2], net.layer_parameters()[1::2])
                       for o, p in ((wo, pw), (bo, pb))]).double()
    assert ((fused - auto).norm() / auto.norm()).item() < 5e-6
    assert loss[0].item() == pytest.approx(((y - t) ** 2).mean().item(), rel=1e-5)
    desc, flat = inr.flat_parameters(net)
    ws = [p.detach().cpu().numpy() for p in net.layer_parameters()[0::2]]
    bs = [p.detach().cpu().numpy() for p in net.layer_parameters()[1::2]]
    want = O.siren_forward(ws, bs, x.cpu().numpy().astype(np.float64), dtype=np.float64)
    got = ops.siren_forward(desc, flat, x).cpu().numpy()
    assert O.rel_l2(got, want) < 1e-5
