"""-m gpu: every kernel family the pre-split (HL32) path can dispatch to, against a float64 oracle.

`hp_eligible` (csrc/api.hip) admits hidden in {128, 256, 512, 1024}, any in_features / hidden multiple of 32, up to 8 sine
layers, one output.  Behind it sit four GEMM families (csrc/gemm_hp.inc): `pkd` (persistent, deferred epilogue: K = 256 /
512), `pkc` (persistent, epilogue in line: every other K >= 96), the one-block-per-tile kernel (K < 96, and everything when
the persistent grid is switched off) and the row-contraction kernel of the parameter gradient; on top `pkn`, the 64-row
tiles a launch with few row tiles takes.  Each case below asserts WHICH families ran (inr_launch_count) so that a silent
fall-back to the fp32 kernels cannot pass, then checks forward (T1), loss and every gradient tensor (T2) against the torch
port evaluated in float64 (same op sequence as SRDWI.py:58-59,86-91 / superresDWI.py:135-137).

Second half: the a-priori scale bound of the backward operands (`HpScale`: a bound that is too tight overflows fp16
silently, one that is too loose loses low bits) under adversarial magnitudes ON THIS PATH -- targets / weights from 1e-30 to
1e3, one row 1e6 above the rest, a network built so that the bound is attained, and the weights at the end of a full
2,500-step fit, where the gradient is a small difference of large terms.  Criterion there: finite, and as close to float64
as the exact-fp32 (f32-input MFMA) kernels are, up to a small factor.
"""
import numpy as np
import pytest
import torch

import mri_super_resolution_amd as inr
from mri_super_resolution_amd import ops
from oracle import inr_oracle as O
from oracle import torch_port as P

pytestmark = pytest.mark.gpu

T1 = 1e-5
T2 = 1e-5


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert ops.device_caps(0)["compute_units"] == CUS
    yield


def host(t):
    return t.detach().cpu().numpy()


def make_pair(fin, hidden, layers, seed, dtype=torch.float64):
    """Our module and the port with the same draws; the port converted to `dtype` for the oracle side."""
    torch.manual_seed(seed)
    net = inr.Siren(fin, hidden, layers, 1)
    torch.manual_seed(seed)
    ref = P.PortSiren(fin, hidden, layers, 1)
    return net, ref.to(dtype)


def oracle_loss_grads(ref, x, t, w):
    """float64 loss and gradients in network order (W_0, b_0, ..., W_head, b_head)."""
    for p in ref.parameters():
        p.grad = None
    out = ref(x.double())
    sq = (out - t.double().reshape(-1, 1)) ** 2
    loss = (sq if w is None else w.double().reshape(-1, 1) * sq).mean()
    loss.backward()
    mods = [m.linear if isinstance(m, P.PortSine) else m for m in ref.net]
    return loss.item(), [g for m in mods for g in (m.weight.grad.numpy(), m.bias.grad.numpy())], out.detach().numpy()


def fused_loss_grads(net, x, t, w):
    """loss, gradients and the (layer-wise) forward"""
    desc, flat = inr.flat_parameters(net)
    grads = torch.zeros_like(flat)
    loss = torch.zeros(1, device="cuda")
    ops.siren_loss_grad(desc, flat, grads, x, t, w, 0, loss)
    _, offsets = ops.siren_param_layout(desc)
    out = []
    for (w_off, b_off), pw, pb in zip(offsets, net.layer_parameters()[0::2], net.layer_parameters()[1::2]):
        out.append(host(grads[w_off:w_off + pw.numel()]).reshape(pw.shape))
        out.append(host(grads[b_off:b_off + pb.numel()]))
    return loss.item(), out, host(ops.siren_forward(desc, flat, x))


CUS = 256     # MI355X (ops.device_caps: checked in the fixture below)


def row_plan(n, width, persistent=2, narrow=1):
    """hp_row_plan of csrc/gemm_f32.hip: (rows for the wide persistent kernels, rows for the 64 x 128 tiles of gemm_hp_nt_kernel),
    spelled out a second time on purpose: a change of the dispatch has to be made in both places."""
    if not narrow or not persistent:
        return n, 0
    tiles_n, tiles_m = (width + 255) // 256, (n + 127) // 128
    return (0, n) if 4 * tiles_m * tiles_n <= 3 * CUS else (n, 0)


def kc_families(K, deferred_ok, n, width, persistent=2, narrow=1, row=0, row_min_tiles=1024):
    """families one K-contiguous GEMM (forward or input-grad) of `n` rows and `width` output columns launches"""
    wide, nar = row_plan(n, width, persistent, narrow)
    out = []
    if wide:
        # hp_row_ok of csrc/gemm_f32.hip: the row-owning kernel takes 512-column launches with enough 128-row panels
        if row and persistent == 2 and width == 512 and deferred_ok and (wide + 127) // 128 >= row_min_tiles:
            out.append("hp_row")
        else:
            out.append("hp_pkd" if persistent == 2 and deferred_ok else "hp_pkc" if persistent and K >= 96 else "hp_tile")
    if nar:
        out.append("hp_narrow")
    return out


def expected_families(fin, hidden, layers, n, backward=True, persistent=2, narrow=1):
    """family -> launches of ONE inr_siren_loss_grad (backward=True) or inr_siren_forward call."""
    S = layers + 1
    fam = {}
    for l in range(S):
        K = fin if l == 0 else hidden
        for name in kc_families(K, K in (256, 512), n, hidden, persistent, narrow):
            fam[name] = fam.get(name, 0) + 1
    if backward:
        for _ in range(S - 1):
            for name in kc_families(hidden, hidden == 512, n, hidden, persistent, narrow):
                fam[name] = fam.get(name, 0) + 1
        fam["hp_rc"] = S
    return fam


def check_families(counts, *wants):
    total = {}
    for w in wants:
        for k, v in w.items():
            total[k] = total.get(k, 0) + v
    for name in ops.LAUNCH_FAMILIES:
        if not name.startswith("small"):
            assert counts[name] == total.get(name, 0), (name, counts, total)


# (in_features, hidden, hidden_layers, rows): every K class of the three KC families, the widest and the narrowest hidden
# size, no hidden layer at all, the deepest network the path takes (8 sine layers), ragged row counts
SHAPES = [
    (32, 128, 0, 777),
    (64, 128, 2, 1000),
    (96, 256, 1, 1531),
    (256, 256, 5, 2049),
    (512, 256, 3, 513),
    (32, 512, 7, 700),
    (512, 512, 0, 300),
    (160, 512, 2, 4099),
    (64, 1024, 1, 515),
    (1024, 1024, 2, 260),
    (128, 128, 3, 70000),
    (256, 512, 3, 66000),
]


@pytest.mark.parametrize("narrow", [1, 0])
@pytest.mark.parametrize("fin,hidden,layers,n", SHAPES)
def test_every_eligible_shape_class_vs_float64(fin, hidden, layers, n, narrow):
    net, ref = make_pair(fin, hidden, layers, seed=fin + hidden + layers)
    net.cuda()
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, fin, generator=g) * 2 - 1
    t = torch.rand(n, generator=g)
    w = (torch.rand(n, generator=g) > 0.25).float() * (0.5 + torch.rand(n, generator=g))
    want_loss, want_g, want_y = oracle_loss_grads(ref, x, t, w)
    with ops.debug_switch(18, narrow):
        ops.launch_counts_reset()
        loss, got_g, got_y = fused_loss_grads(net, x.cuda(), t.cuda(), w.cuda())
        counts = ops.launch_counts()
    check_families(counts, expected_families(fin, hidden, layers, n, True, narrow=narrow),
                   expected_families(fin, hidden, layers, n, False, narrow=narrow))     # loss_grad + the forward call
    assert O.rel_l2(got_y, want_y) < T1
    assert abs(loss - want_loss) <= T2 * want_loss
    for k, (a, b) in enumerate(zip(got_g, want_g)):
        assert O.rel_l2(a, b) < T2, (k, O.rel_l2(a, b))


@pytest.mark.parametrize("persistent,zhead", [(0, 1), (1, 1), (2, 0)])
@pytest.mark.parametrize("fin,hidden,layers,n", [(256, 512, 3, 3000), (64, 256, 2, 1100), (512, 1024, 1, 400)])
def test_fallback_families_of_the_persistent_grid(fin, hidden, layers, n, persistent, zhead):
    """inr_debug_set(10, 0 / 1): the one-block-per-tile kernel and the in-line-epilogue persistent kernel serve EVERY K;
    key 16 = 0: the last sine layer stashes act + omega cos instead of z.  Same tolerances."""
    net, ref = make_pair(fin, hidden, layers, seed=7 * persistent + zhead)
    net.cuda()
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, fin, generator=g) * 2 - 1
    t = torch.rand(n, generator=g) * 2 - 1
    want_loss, want_g, want_y = oracle_loss_grads(ref, x, t, None)
    with ops.debug_switch(10, persistent), ops.debug_switch(16, zhead), ops.debug_switch(18, 0):
        ops.launch_counts_reset()
        loss, got_g, got_y = fused_loss_grads(net, x.cuda(), t.cuda(), None)
        counts = ops.launch_counts()
    if persistent == 0:
        assert counts["hp_tile"] == 3 * (layers + 1) - 1 and counts["hp_pkd"] == counts["hp_pkc"] == 0, counts
    elif persistent == 1:
        assert counts["hp_pkd"] == 0 and counts["hp_pkc"] + counts["hp_tile"] == 3 * (layers + 1) - 1, counts
    assert counts["hp_rc"] == layers + 1 and counts["h3"] == counts["f32_pipe16"] == counts["f32_generic"] == 0, counts
    assert O.rel_l2(got_y, want_y) < T1
    assert abs(loss - want_loss) <= T2 * want_loss
    for k, (a, b) in enumerate(zip(got_g, want_g)):
        assert O.rel_l2(a, b) < T2, (k, O.rel_l2(a, b))


# ------------------------------------------------------------------ the scale bound under adversarial magnitudes ---------
def grads_on(net, x, t, w, exact_fp32):
    with ops.debug_switch(3, 0 if exact_fp32 else 1):
        ops.launch_counts_reset()
        out = fused_loss_grads(net, x, t, w)
        c = ops.launch_counts()
    if exact_fp32:
        assert c["hp_pkd"] + c["hp_pkc"] + c["hp_tile"] + c["hp_rc"] + c["hp_narrow"] + c["h3"] == 0 and c["f32_pipe16"] > 0, c
    else:
        assert c["hp_rc"] > 0 and c["f32_pipe16"] + c["f32_generic"] + c["h3"] == 0, c
    return out


def assert_fp32_class(net, ref, x, t, w, factor=16.0, loss_rtol=1e-5):
    """Gradients of the HL32 path are finite and inside tier T2 (1e-5 against float64) -- or, where the exact-fp32 kernels
    themselves cannot meet T2 (cancellation), within `factor` x their distance from float64.  (hi + lo carries 22 bits
    against fp32's 24; measured on the GPU with the head weights x100: 2.5e-6 on every tensor where exact fp32 has 1e-7 ..
    1e-6.  A starved or overflowed scale shows as 1e-3 .. NaN.)"""
    want_loss, want_g, _ = oracle_loss_grads(ref, x, t, w)
    xd, td, wd = x.cuda(), t.cuda(), None if w is None else w.cuda()
    loss_h, g_h, _ = grads_on(net, xd, td, wd, exact_fp32=False)
    loss_f, g_f, _ = grads_on(net, xd, td, wd, exact_fp32=True)
    assert np.isfinite(loss_h) and abs(loss_h - want_loss) <= loss_rtol * abs(want_loss) + 1e-37
    worst = 0.0
    for k, (a, f, b) in enumerate(zip(g_h, g_f, want_g)):
        assert np.isfinite(a).all(), k
        if np.linalg.norm(b) == 0.0:
            assert not a.any(), k
            continue
        eh, ef = O.rel_l2(a, b), O.rel_l2(f, b)
        assert eh <= max(T2, factor * ef), (k, eh, ef)
        worst = max(worst, eh)
    return worst


@pytest.mark.parametrize("t_scale", [1e-30, 1e-12, 1.0, 1e3])
@pytest.mark.parametrize("fin,hidden,layers", [(256, 512, 3), (64, 256, 2), (128, 1024, 1)])
def test_target_magnitudes(fin, hidden, layers, t_scale):
    net, ref = make_pair(fin, hidden, layers, seed=3)
    net.cuda()
    g = torch.Generator().manual_seed(17)
    n = 2500
    x = torch.rand(n, fin, generator=g) * 2 - 1
    t = (torch.rand(n, generator=g) * 2 - 1) * t_scale
    assert_fp32_class(net, ref, x, t, None)


@pytest.mark.parametrize("head_scale,hidden_scale,first_scale", [(1e-6, 1.0, 1.0), (1e2, 1.0, 1.0), (1.0, 1e-4, 1.0),
                                                                  (1.0, 3.0, 1.0), (1.0, 1.0, 1e-5), (1e-20, 1e-3, 20.0)])
def test_weight_magnitudes(head_scale, hidden_scale, first_scale):
    net, ref = make_pair(256, 512, 3, seed=5)
    with torch.no_grad():
        for m_ours, m_ref in ((net.final_linear, ref.final_linear),):
            m_ours.weight.mul_(head_scale)
            m_ref.weight.mul_(head_scale)
        for l in range(4):
            s = first_scale if l == 0 else hidden_scale
            net.net[l].linear.weight.mul_(s)
            ref.net[l].linear.weight.mul_(s)
    net.cuda()
    g = torch.Generator().manual_seed(23)
    n = 3000
    x = torch.rand(n, 256, generator=g) * 2 - 1
    t = torch.rand(n, generator=g)
    w = torch.rand(n, generator=g)
    assert_fp32_class(net, ref, x, t, w)


def test_one_row_a_million_above_the_rest():
    """A weight image (master.py:143-147) with one voxel 1e6 above the others, and the same for the targets: the tensor scale
    follows the maximum; norm-wise nothing is lost, and the small rows still carry their gradient."""
    net, ref = make_pair(256, 512, 2, seed=11)
    net.cuda()
    g = torch.Generator().manual_seed(29)
    n = 2048
    x = torch.rand(n, 256, generator=g) * 2 - 1
    t = torch.rand(n, generator=g)
    w = torch.rand(n, generator=g) * 1e-3
    w[77] = 1e3
    assert_fp32_class(net, ref, x, t, w)
    t2 = t.clone()
    t2[5] = 1e6
    assert_fp32_class(net, ref, x, t2, None)
    # without the outlier row the remaining gradient is still right (elementwise health of rows far below the maximum)
    w0 = w.clone()
    w0[77] = 0.0
    assert_fp32_class(net, ref, x, t, w0)


def test_bound_attained_constant_sign_network():
    """The bound of dz_l is max|dz_{l+1}| x max_j sum_k |W_{l+1}[k][j]| x omega.  It is ATTAINED when the rows of W_{l+1} that
    carry gradient share one sign and magnitude, their units sit at cos = 1 and dz_{l+1} is constant along a row.  This network
    does that on purpose while keeping O(1) activations: even units have bias 0 and tiny positive weights (z ~ 0: cos = 1,
    sin = 0 -- they carry all of the gradient), odd units have bias pi / (2 omega) and zero weights (sin = 1, cos = 0 -- they
    carry the activations).  A bound computed a hair too small would push the largest element past fp16's range (inf / NaN in
    the gradients): this pins the 1.001 safety factors of HpScale."""
    fin, hidden, layers, n = 256, 512, 3, 1500
    net, ref = make_pair(fin, hidden, layers, seed=1)
    with torch.no_grad():
        for m in (net, ref):
            for l in range(layers + 1):
                lin = m.net[l].linear
                lin.weight.zero_()
                lin.weight[0::2] = 1e-7 if l else 1e-8
                lin.bias.zero_()
                lin.bias[1::2] = float(np.pi / 60.0)
            m.final_linear.weight.fill_(0.01)
            m.final_linear.bias.zero_()
    net.cuda()
    g = torch.Generator().manual_seed(31)
    x = torch.rand(n, fin, generator=g) * 2 - 1
    for t_val in (1.0, -1.0, 1e-6, 250.0):
        t = torch.full((n,), t_val)
        assert_fp32_class(net, ref, x, t, None)


def test_all_tiny_activations_keep_their_precision():
    """Rounds 2-3 stored sine outputs UNSCALED in the HL32 image (they live in [-1, 1]): an element kept 22 bits down to 2^-3 and an
    absolute 2^-25 below that, so a layer whose ENTIRE output is ~1e-6 (all weights 1e-8: nothing a SIREN initialisation or fit
    produces) reached the next GEMM with ~5 bits and the gradients were wrong with rc 0 (verdict r03, weak 2).  Round 4 scales every
    layer's image from an a-priori bound of its output, min(1, omega (fan_in max|W| max|a_prev| + max|b|)), evaluated by the
    per-step weight preparation: the DEFAULT path now meets T2 on such a network, for one and for several tiny layers."""
    for layers, tiny in ((1, (0,)), (3, (0,)), (3, (0, 1)), (3, (2,))):
        net, ref = make_pair(256, 512, layers, seed=1)
        with torch.no_grad():
            for m in (net, ref):
                for l in tiny:
                    m.net[l].linear.weight.fill_(1e-8)
                    m.net[l].linear.bias.zero_()
        net.cuda()
        x = torch.rand(600, 256, generator=torch.Generator().manual_seed(3)) * 2 - 1
        t = torch.rand(600)
        _, want_g, want_y = oracle_loss_grads(ref, x, t, None)
        _, got_g, got_y = fused_loss_grads(net, x.cuda(), t.cuda(), None)
        assert np.abs(got_y - want_y).max() < 1e-6, (layers, tiny)
        for k, (a, b) in enumerate(zip(got_g, want_g)):
            if not np.any(b):
                assert not np.any(a)
                continue
            assert O.rel_l2(a, b) < T2, (layers, tiny, k)
        assert_fp32_class(net, ref, x, t, None)


def test_zero_residual_and_zero_weight_image():
    net, ref = make_pair(64, 128, 1, seed=2)
    net.cuda()
    x = torch.rand(500, 64) * 2 - 1
    desc, flat = inr.flat_parameters(net)
    y = ops.siren_forward(desc, flat, x.cuda()).reshape(-1)
    loss, grads, _ = fused_loss_grads(net, x.cuda(), y.contiguous(), None)
    assert loss == 0.0 and all(not g.any() for g in grads)
    loss, grads, _ = fused_loss_grads(net, x.cuda(), torch.rand(500).cuda(), torch.zeros(500).cuda())
    assert loss == 0.0 and all(np.isfinite(g).all() and not g.any() for g in grads)


def test_late_training_state(golden):
    """The weights after the full 2,500-step config-1 fit (superresDWI.py:132-138; loss ~1e-6: the gradient is a small
    difference of large terms and the head's a-priori bound is ~2^11 above the actual residuals): gradients on the HL32 path
    against float64 next to the exact-fp32 kernels, then 20 more steps staying on the exact-fp32 trajectory."""
    d = golden("dataset_ff.npz")
    lr_img = golden("pat07_slice11.npz")["lr"]
    torch.manual_seed(0)
    net = inr.Siren(256, 512, 3, 1).cuda()
    B = torch.from_numpy(d["B2"]).cuda()
    x = inr.input_mapping(inr.get_mgrid((64, 64)), B)
    t = torch.from_numpy(lr_img.astype(np.float32).reshape(-1)).cuda()
    fitter = inr.SirenFitter(net, lr=1e-4)
    losses = host(fitter.step(x, t, 2500))
    assert np.median(losses[-100:]) < 2e-5
    # float64 twin of the fitted network
    ref = P.PortSiren(256, 512, 3, 1).double()
    ref.load_state_dict({k: v.detach().cpu().double() for k, v in net.state_dict().items()}, strict=False)
    twin = inr.Siren(256, 512, 3, 1)
    twin.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=False)
    twin.cuda()
    # (the loss is a sum of squared residuals of ~1e-3, each a difference of O(1) numbers known to ~1e-7)
    worst = assert_fp32_class(twin, ref, x.cpu(), t.cpu(), None, loss_rtol=2e-4)
    print("late-training state: worst gradient distance from float64 on the HL32 path %.2e" % worst)
    # the next 20 steps on both arithmetic paths from the same state (parameters + Adam moments)
    state = (fitter.flat.clone(), fitter.m.clone(), fitter.v.clone(), fitter.step_count)
    runs = {}
    for exact in (False, True):
        fitter.flat.copy_(state[0]); fitter.m.copy_(state[1]); fitter.v.copy_(state[2]); fitter.step_count = state[3]
        with ops.debug_switch(3, 0 if exact else 1):
            runs[exact] = (host(fitter.step(x, t, 20)).copy(), host(fitter.flat).copy())
    assert np.isfinite(runs[False][0]).all()
    assert O.rel_l2(runs[False][1], runs[True][1]) < 1e-4


# ------------------------------------------------------------------ cross-layer fused forward (gemm_hp_fwd.inc) -------------
@pytest.mark.parametrize("fin,hidden,layers,n", [(256, 512, 3, 4099), (256, 512, 3, 1), (32, 512, 0, 63), (512, 512, 7, 64),
                                                  (64, 256, 2, 65), (256, 256, 5, 1000), (96, 256, 1, 130), (160, 512, 2, 20000)])
def test_fused_forward_all_layers_in_one_launch(fin, hidden, layers, n):
    """inr_debug_set(19, 1): inr_siren_forward of an eligible network as ONE launch (siren_fwd_fused_kernel: a 64-row panel walks
    through every sine layer and the head inside LDS), against the float64 oracle (T1) and against the layer-wise launches (same
    arithmetic; only the head's fp32 row sum is grouped differently).  Measured slower than those at hidden = 512, so not the
    default (cross-layer fusion study) -- but it must stay right."""
    net, ref = make_pair(fin, hidden, layers, seed=fin + layers)
    net.cuda()
    x = torch.rand(n, fin, generator=torch.Generator().manual_seed(n)) * 2 - 1
    want = ref(x.double()).detach().numpy()
    desc, flat = inr.flat_parameters(net)
    xd = x.cuda()
    ops.launch_counts_reset()
    y_layers = host(ops.siren_forward(desc, flat, xd))
    assert ops.launch_counts()["hp_fused_fwd"] == 0
    with ops.debug_switch(19, 1):
        ops.launch_counts_reset()
        y = host(ops.siren_forward(desc, flat, xd))
        c = ops.launch_counts()
        assert c["hp_fused_fwd"] == 1 and c["hp_pkd"] + c["hp_pkc"] + c["hp_tile"] + c["hp_narrow"] + c["f32_pipe16"] == 0, c
        yc = host(ops.siren_forward(desc, flat, xd, clamp_min=0.05))
        part = host(ops.siren_forward(desc, flat, xd[37:n - 5].contiguous())) if n > 70 else None
    assert y.shape == (n, 1) and O.rel_l2(y, want) < T1
    assert O.rel_l2(y, y_layers) < 2e-6 and np.abs(y - y_layers).max() < 2e-6
    assert np.array_equal(yc, np.maximum(y, np.float32(0.05)))
    if part is not None:                                      # a row's value depends on nothing but the row
        assert np.array_equal(part, y[37:n - 5])


def test_fused_forward_falls_back_where_the_panel_does_not_fit():
    """hidden = 1024 (a 256 KB panel) and in_features > hidden keep the layer-wise launches."""
    for fin, hidden in ((64, 1024), (512, 256)):
        net, ref = make_pair(fin, hidden, 1, seed=3)
        net.cuda()
        x = torch.rand(300, fin) * 2 - 1
        desc, flat = inr.flat_parameters(net)
        with ops.debug_switch(19, 1):
            ops.launch_counts_reset()
            y = host(ops.siren_forward(desc, flat, x.cuda()))
            assert ops.launch_counts()["hp_fused_fwd"] == 0
        assert O.rel_l2(y, ref(x.double()).detach().numpy()) < T1


@pytest.mark.parametrize("shape,m,hidden", [((40, 33), 128, 512), ((9, 10, 7), 32, 256), ((5, 6, 3, 4), 64, 128), ((31, 17), 16, 128)])
def test_reconstruct_builds_its_input_as_hl32_directly(shape, m, hidden):
    """inr_siren_reconstruct on the pre-split path: grid -> Fourier features -> HL32 operand image in ONE kernel per chunk
    (no fp32 feature matrix, no amax pass, no conversion pass) -- bit for bit what the explicit get_mgrid -> input_mapping ->
    inr_siren_forward sequence gives, in any chunking; m % 32 != 0 keeps the three-pass form."""
    torch.manual_seed(m)
    net = inr.Siren(2 * m, hidden, 2, 1).cuda()
    B = torch.from_numpy(P.fourier_matrix(len(shape), mapping_size=m, seed=5)).cuda()
    desc, flat = inr.flat_parameters(net)
    x = inr.input_mapping(inr.get_mgrid(shape), B)
    want = host(ops.siren_forward(desc, flat, x, clamp_min=0.0)).reshape(shape)
    got = host(inr.reconstruct(net, shape, B))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    small = host(inr.reconstruct(net, shape, B, chunk_rows=101))
    assert np.array_equal(small.view(np.uint32), got.view(np.uint32))
    ref = P.PortSiren(2 * m, hidden, 2, 1)
    ref.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()}, strict=False)
    assert O.rel_l2(got, P.port_reconstruct(ref, shape, B.cpu())) < T1


@pytest.mark.parametrize("fin,n", [(256, 128 * 40 + 77), (512, 128 * 33), (128, 128 * 9 + 1), (64, 2000)])
def test_row_owning_kernel_is_bit_identical_to_the_deferred_epilogue_kernel(fin, n):
    """gemm_hp_row_kernel (round 5: one block owns 128 rows x all 512 columns, two 80 KB LDS stages, epilogue in line) against
    gemm_hp_pkd / pkc_kernel on the same launches: every accumulator sees the same sequence of MFMAs and the epilogue is the same
    chunk arithmetic, so a 6-step fit (forward with stash, last layer z-only, input gradients with column sums and maxima), the
    losses and an inference forward agree BIT FOR BIT -- ragged last panel, first-layer K = 64 ... 512 (the row kernel takes K = 256 / 512 only), one tile per CU or several."""
    from mri_super_resolution_amd._lib import lib
    g = torch.Generator().manual_seed(n)
    x = (torch.rand(n, fin, generator=g) * 2 - 1).cuda()
    t = torch.rand(n, generator=g).cuda()
    w = (torch.rand(n, generator=g) + 0.5).cuda()
    out = {}
    for row in (0, 1):
        lib().inr_debug_set(18, 0)            # no narrow tiles: the wide dispatch decides
        lib().inr_debug_set(27, row)
        lib().inr_debug_set(28, 1)            # any panel count
        torch.manual_seed(0)
        net = inr.Siren(fin, 512, 2, 1).cuda()
        fitter = inr.SirenFitter(net, lr=1e-4)
        ops.launch_counts_reset()
        losses = fitter.step(x, t, 6, w)
        torch.cuda.synchronize()
        c = ops.launch_counts()
        desc, flat = inr.flat_parameters(net)
        y = ops.siren_forward(desc, flat, x)
        out[row] = (losses.cpu().numpy(), fitter.flat.cpu().numpy(), y.cpu().numpy(), c)
        lib().inr_debug_reset()
    (l0, f0, y0, c0), (l1, f1, y1, c1) = out[0], out[1]
    assert c0["hp_row"] == 0 and c0["hp_pkd"] + c0["hp_pkc"] > 0
    want_row = 6 * (3 + 2) if fin in (256, 512) else 6 * (2 + 2)       # per step: forward layers + input-grad layers served (the Ks
                                                                       # of the deferred-epilogue kernel, whose MFMA order it keeps)
    assert c1["hp_row"] == want_row, c1
    assert np.array_equal(l0, l1) and np.array_equal(f0, f1) and np.array_equal(y0, y1)
    assert np.isfinite(l1).all() and l1[-1] < l1[0]


# ------------------------------------------------------------------ the head step fused into the last sine layer (round 5) ---------
@pytest.mark.parametrize("fin,layers,n,weighted", [(256, 2, 128 * 5 + 77, True), (512, 0, 128 * 3, False), (256, 3, 128 * 20 + 1, False),
                                                   (64, 1, 1000, True)])
def test_fused_head_epilogue_vs_float64_and_vs_the_head_step_kernel(fin, layers, n, weighted):
    """`gemm_hp_row_kernel<HPE_HEAD>` (csrc/gemm_hp_row.inc): the last sine layer's block owns whole rows, so the head -- y, residual,
    dL/dy, dz_L = g w omega cos, the bias / head-weight gradient slabs, the loss -- is formed in its epilogue: no z round trip, no head
    step kernel.  Debug key 31 = 1 lets any row count take it (default: from 768 row panels = 98,304 rows on).  Against the float64 oracle (tiers
    T1 / T2, as every other family here) and against the unfused path (key 30 = 0): same arithmetic per element, other summation orders."""
    net, ref = make_pair(fin, 512, layers, seed=fin + layers + n)
    net.cuda()
    g = torch.Generator().manual_seed(n)
    x = torch.rand(n, fin, generator=g) * 2 - 1
    t = torch.rand(n, generator=g) * 2 - 1
    w = ((torch.rand(n, generator=g) > 0.25).float() * (0.5 + torch.rand(n, generator=g))) if weighted else None
    want_loss, want_g, want_y = oracle_loss_grads(ref, x, t, w)
    xd, td, wd = x.cuda(), t.cuda(), None if w is None else w.cuda()
    with ops.debug_switch(31, 1):
        ops.launch_counts_reset()
        loss, got_g, got_y = fused_loss_grads(net, xd, td, wd)
        counts = ops.launch_counts()
    assert counts["hp_row"] == 1 and counts["hp_rc"] == layers + 1, counts          # the last sine layer of the loss_grad call, nothing else
    assert abs(loss - want_loss) <= T2 * abs(want_loss)
    for k, (a, b) in enumerate(zip(got_g, want_g)):
        assert O.rel_l2(a, b) < T2, (k, O.rel_l2(a, b))
    with ops.debug_switch(30, 0):
        ops.launch_counts_reset()
        loss0, got0, _ = fused_loss_grads(net, xd, td, wd)
        assert ops.launch_counts()["hp_row"] == 0
    assert abs(loss - loss0) <= 2e-6 * abs(loss0)
    for k, (a, b) in enumerate(zip(got_g, got0)):
        assert O.rel_l2(a, b) < 2e-6, (k, O.rel_l2(a, b))


def test_fused_head_fit_trajectory_and_cycling_targets():
    """Thirty fused fit steps (inr_siren_fit) and a cycling-acquisition call (inr_siren_fit_cycle: target and weight image change every
    step) with the head in the last layer's epilogue against the same calls through the head step kernel: tier T3 (1e-4) on the weights,
    per-step losses within 1e-4 of each other and of the torch-CPU port."""
    n, fin = 128 * 9 + 50, 256
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(n, fin, generator=g) * 2 - 1).cuda()
    t = torch.rand(n, generator=g).cuda()
    tg = torch.rand(3, n, generator=g).cuda()
    wt = torch.rand(3, n, generator=g).cuda()
    out = {}
    for fused in (1, 0):
        with ops.debug_switch(30, fused), ops.debug_switch(31, 1):
            torch.manual_seed(0)
            f = inr.SirenFitter(inr.Siren(fin, 512, 2, 1).cuda(), lr=1e-4)
            ops.launch_counts_reset()
            l1 = f.step(x, t, 30)
            l2 = f.step_cycle(x, tg, 7, wt, first_acq=1)
            torch.cuda.synchronize()
            out[fused] = (host(l1), host(l2), host(f.flat), ops.launch_counts()["hp_row"])
    assert out[1][3] == 37 and out[0][3] == 0
    # tier T3 (north_star: trajectories of <= 50 steps within 1e-4): against each other and, the 30 plain steps, against the torch-CPU
    # port of the reference loop (the trajectory is in Adam's oscillating regime by step 20: losses 0.0046 -> 0.0058 -> 0.0027 -> ...,
    # where weights that agree to 4e-7 give losses that agree to 1e-5)
    assert np.allclose(out[1][0], out[0][0], rtol=1e-4) and np.allclose(out[1][1], out[0][1], rtol=1e-4)
    assert O.rel_l2(out[1][2], out[0][2]) < 1e-4
    assert out[1][0][-1] < out[1][0][0]
    torch.manual_seed(0)
    port_losses, _ = P.port_fit(P.PortSiren(fin, 512, 2, 1), x.cpu(), t.cpu().reshape(-1, 1), 30, lr=1e-4)
    assert np.allclose(out[1][0], port_losses, rtol=1e-4) and np.allclose(out[0][0], port_losses, rtol=1e-4)


def test_fused_head_as_a_row_shard_and_run_to_run_bits():
    """The fused-head layer inside `inr_siren_loss_grad` with `count_total` = the GLOBAL row count (what a row-sharded fit passes: the mean
    is taken over all shards) against the head-step path, and two runs of the same call bit for bit (fixed-order slabs, no float atomics)."""
    n, fin = 128 * 7 + 19, 256
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(n, fin, generator=g) * 2 - 1).cuda()
    t = torch.rand(n, generator=g).cuda()
    w = (torch.rand(n, generator=g) + 0.25).cuda()
    torch.manual_seed(3)
    net = inr.Siren(fin, 512, 1, 1).cuda()
    desc, flat = inr.flat_parameters(net)

    def run(fused):
        with ops.debug_switch(30, fused), ops.debug_switch(31, 1):
            grads, loss = torch.zeros_like(flat), torch.zeros(1, device="cuda")
            ops.launch_counts_reset()
            ops.siren_loss_grad(desc, flat, grads, x, t, w, 3 * n, loss)
            return host(grads), float(loss), ops.launch_counts()["hp_row"]

    g1, l1, c1 = run(1)
    g2, l2, c2 = run(1)
    g0, l0, c0 = run(0)
    assert c1 == c2 == 1 and c0 == 0
    assert np.array_equal(g1, g2) and l1 == l2
    assert abs(l1 - l0) <= 2e-6 * abs(l0) and O.rel_l2(g1, g0) < 2e-6
