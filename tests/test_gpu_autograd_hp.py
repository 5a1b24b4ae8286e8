"""-m gpu: the reference's OWN loop (superresDWI.py:132-138: ``INR(x)`` -> torch forms the loss -> ``loss.backward()`` ->
``torch.optim.Adam.step()``) on the fused fit's kernels (``_SirenHpFn``: inr_siren_forward_train / inr_siren_backward_train)
against the reference fixtures, the torch-CPU port and the layer-by-layer exact-fp32 path it replaces."""
import numpy as np
import pytest
import torch

import mri_super_resolution_amd as inr
from mri_super_resolution_amd import inr as inr_mod
from mri_super_resolution_amd import ops
from oracle import inr_oracle as O
from oracle import torch_port as P
from tests.conftest import strided_sample

pytestmark = pytest.mark.gpu
T1 = T2 = 1e-5
T3 = 1e-4


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture
def legacy_autograd():
    inr_mod.HP_AUTOGRAD = False
    yield
    inr_mod.HP_AUTOGRAD = True


def _net(golden, flavor="SRDWI"):
    d = golden("dataset_ff.npz")
    torch.manual_seed(0)
    net = inr.Siren(256, 512, 3, 1, flavor=flavor).cuda()
    x = inr.input_mapping(inr.get_mgrid((64, 64)), dev(d["B2"]))
    return net, x, d


def test_which_kernels_each_mode_runs(golden):
    net, x, d = _net(golden)
    t = dev(d["lr_pixels"][0])
    ops.launch_counts_reset()
    ((net(x) - t) ** 2).mean().backward()
    c = ops.launch_counts()
    assert c["hp_narrow"] + c["hp_pkd"] + c["hp_pkc"] + c["hp_tile"] >= 7 and c["hp_rc"] == 4 and c["f32_pipe16"] == 0, c
    inr_mod.HP_AUTOGRAD = False
    try:
        net.zero_grad()
        ops.launch_counts_reset()
        ((net(x) - t) ** 2).mean().backward()
        c = ops.launch_counts()
        assert c["f32_pipe16"] >= 11 and c["hp_rc"] == 0, c
    finally:
        inr_mod.HP_AUTOGRAD = True
    with torch.no_grad():                                   # inference under no_grad never takes the stash path
        ops.launch_counts_reset()
        net(x)
        assert ops.launch_counts()["hp_rc"] == 0


@pytest.mark.parametrize("flavor", ["SRDWI", "INRmodel"])
def test_forward_and_gradients_match_reference(golden, flavor):
    g = golden("siren512_step0.npz")
    net, x, d = _net(golden, flavor)
    t = dev(d["lr_pixels"][0])
    out = net(x)
    assert O.rel_l2(host(out), g[f"{flavor}/fwd"]) < T1
    loss = ((out - t) ** 2).mean()
    loss.backward()
    if flavor == "SRDWI":
        assert abs(loss.item() - g["SRDWI/loss0"]) / g["SRDWI/loss0"] < T2
        for n, p in net.named_parameters():
            gr = host(p.grad)
            assert O.rel_l2(strided_sample(gr), g[f"SRDWI/grad_strided/{n}"]) < T2, n
            nrm = np.linalg.norm(gr.astype(np.float64))
            assert abs(nrm - g[f"SRDWI/grad_norm/{n}"]) / g[f"SRDWI/grad_norm/{n}"] < T2, n
    torch.manual_seed(0)
    ref = P.PortSiren(256, 512, 3, 1, flavor=flavor)
    ((ref(torch.from_numpy(host(x))) - torch.from_numpy(d["lr_pixels"][0])) ** 2).mean().backward()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert O.rel_l2(host(p.grad), q.grad.numpy()) < T2, n


def test_arbitrary_upstream_gradient_and_weighted_loss_vs_float64(golden):
    """dL/dy is whatever autograd hands over (here a weighted, cubed residual): every gradient tensor against float64."""
    net, x, d = _net(golden)
    t = dev(d["lr_pixels"][0])
    w = dev(np.random.default_rng(3).random((4096, 1)) + 0.1)
    (w * (net(x) - t).abs() ** 3).sum().backward()
    torch.manual_seed(0)
    ref = P.PortSiren(256, 512, 3, 1).double()
    (torch.from_numpy(host(w)).double() * (ref(torch.from_numpy(host(x)).double()) - torch.from_numpy(d["lr_pixels"][0]).double()).abs() ** 3
     ).sum().backward()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert O.rel_l2(host(p.grad), q.grad.numpy()) < T2, n


def test_matches_the_layer_by_layer_path_and_the_fused_fit(golden):
    net, x, d = _net(golden)
    t = dev(d["lr_pixels"][0])
    ((net(x) - t) ** 2).mean().backward()
    hp = {n: host(p.grad).copy() for n, p in net.named_parameters()}
    net.zero_grad()
    inr_mod.HP_AUTOGRAD = False
    try:
        ((net(x) - t) ** 2).mean().backward()
    finally:
        inr_mod.HP_AUTOGRAD = True
    for n, p in net.named_parameters():
        assert O.rel_l2(hp[n], host(p.grad)) < 3e-6, n
    fitter = inr.SirenFitter(net, lr=0.0)
    fitter.step(x, t, n_steps=1)
    names = ["net.%d.linear.%s" % (l, k) for l in range(4) for k in ("weight", "bias")] + ["final_linear.weight", "final_linear.bias"]
    for l, (w_off, b_off) in enumerate(fitter.offsets):
        gw = host(fitter.grads[w_off:w_off + hp[names[2 * l]].size]).reshape(hp[names[2 * l]].shape)
        assert O.rel_l2(gw, hp[names[2 * l]]) < 1e-6, names[2 * l]        # same kernels; only dL/dy is formed elsewhere


def test_accumulation_pending_forwards_and_inplace_inputs(golden, monkeypatch):
    net, x, d = _net(golden)
    t = dev(d["lr_pixels"][0])
    flags_seen = []
    real_forward_train = ops.siren_forward_train

    def spy(desc, flat, xx, ws, flags=0):
        flags_seen.append(int(flags))
        return real_forward_train(desc, flat, xx, ws, flags)

    monkeypatch.setattr(ops, "siren_forward_train", spy)
    ((net(x) - t) ** 2).mean().backward()
    once = {n: p.grad.clone() for n, p in net.named_parameters()}
    ((net(x) - t) ** 2).mean().backward()                                      # accumulates into .grad
    for n, p in net.named_parameters():
        assert torch.allclose(p.grad, 2 * once[n], rtol=1e-6, atol=0), n
    # the loop of the reference hands the SAME input every step (through a fresh .detach() view): its operand image is reused
    assert flags_seen == [0, ops.REUSE_INPUT_IMAGE], flags_seen
    net.zero_grad()
    x2 = x.flip(0).contiguous()
    y1, y2 = net(x), net(x2)                                                   # two forwards pending, backward through both
    (((y1 - t) ** 2).mean() + ((y2 - t.flip(0)) ** 2).mean()).backward()
    for n, p in net.named_parameters():
        assert O.rel_l2(host(p.grad), host(2 * once[n])) < 2e-6, n
    net.zero_grad()
    xm = x.clone()
    a = net(xm)
    xm.mul_(0.5)                                                               # same storage, new contents: no operand-image reuse
    b = net(xm)
    with torch.no_grad():
        want = net(x * 0.5)
    assert O.rel_l2(host(b), host(want)) < 1e-6 and O.rel_l2(host(a), host(b)) > 1e-2
    (b.sum()).backward()
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())
    # a NEW batch tensor at the address (and version counter) of the one the previous step freed -- what a mini-batch loop does --
    # is not the tensor whose operand image the workspace holds
    net.zero_grad()
    outs, ptrs = [], []
    for k in range(3):
        xb = x * (1.0 - 0.25 * k)                                              # fresh tensor each step, freed after its backward
        ptrs.append(xb.data_ptr())
        yb = net(xb)
        outs.append(yb.detach().clone())
        yb.sum().backward()
        del xb, yb
    assert flags_seen[-3:] == [0, 0, 0], flags_seen          # (no batch was taken for the previous one)
    with torch.no_grad():
        for k in range(3):
            assert O.rel_l2(host(outs[k]), host(net(x * (1.0 - 0.25 * k)))) < 1e-6, (k, ptrs)
    assert O.rel_l2(host(outs[0]), host(outs[1])) > 1e-2


def test_reference_loop_verbatim_ten_steps(golden):
    """superresDWI.py:132-138 as written, 10 steps, against the reference trajectory (T3) -- on both autograd paths."""
    tr = golden("siren512_traj.npz")
    for hp in (True, False):
        inr_mod.HP_AUTOGRAD = hp
        try:
            INR, model_input, d = _net(golden)
            ground_truth = dev(d["lr_pixels"][0])
            inr_optim = torch.optim.Adam(lr=1e-4, params=INR.parameters())
            for ctr in range(10):
                model_output = INR(model_input)
                loss = ((model_output - ground_truth) ** 2).mean()
                inr_optim.zero_grad()
                loss.backward()
                inr_optim.step()
            rec = host(inr.reconstruct(INR, (128, 128), dev(d["B2"])))
            assert O.rel_l2(rec, tr["t8/recon_10"]) < T3, hp
        finally:
            inr_mod.HP_AUTOGRAD = True


def test_small_and_ineligible_shapes_keep_working():
    torch.manual_seed(1)
    net = inr.Siren(2, 64, 6, 1).cuda()                                        # not served by the pre-split kernels' entry points?
    x = torch.rand(300, 2, device="cuda") * 2 - 1
    net(x).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
    torch.manual_seed(1)
    ref = P.PortSiren(2, 64, 6, 1)
    ref(x.cpu()).sum().backward()
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        assert O.rel_l2(host(p.grad), q.grad.numpy()) < 5e-5, n


def test_dropped_graph_does_not_leave_a_stale_operand_image(golden):
    """a training-mode forward whose graph is dropped WITHOUT a backward never gives its stash workspace back; the
    caching allocator may hand that block to other tensors and then to the next forward's workspace again -- same address, same x,
    same n, but the operand image of x in it has been overwritten.  The reuse of the image is only trusted on a workspace the
    state kept, so the second forward must equal a fresh network's forward bit for bit."""
    net, x, d = _net(golden)
    fresh, _, _ = _net(golden)
    want = host(fresh(x))
    y0 = net(x)                                     # graph 1: forward only
    first = host(y0)
    del y0                                          # dropped: its workspace is freed, not given back
    need = ops.siren_fit_workspace_bytes(net.desc(), x.shape[0])
    junk = torch.full((need,), 0x5A, dtype=torch.uint8, device="cuda")       # very likely the block the workspace lived in
    junk2 = torch.full((need,), 0xA5, dtype=torch.uint8, device="cuda")
    del junk, junk2
    second = host(net(x))
    assert np.array_equal(first, want) and np.array_equal(second, want)
    # and the normal cycle (forward, backward, forward) still reuses its workspace and stays bit-identical
    net.zero_grad()
    ((net(x) - 0.5) ** 2).mean().backward()
    net.zero_grad()
    assert np.array_equal(host(net(x)), want)


def test_second_backward_on_one_forward_says_what_to_do(golden):
    net, x, d = _net(golden)
    y = net(x)
    y.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="ONE backward per forward"):
        y.sum().backward()
    inr_mod.HP_AUTOGRAD = False                     # the layer-by-layer path supports it
    try:
        net.zero_grad()
        y = net(x)
        y.sum().backward(retain_graph=True)
        g1 = [p.grad.clone() for p in net.parameters()]
        y.sum().backward()
        assert all(torch.allclose(p.grad, 2 * g) for p, g in zip(net.parameters(), g1))
    finally:
        inr_mod.HP_AUTOGRAD = True


def test_switch_flipped_after_the_first_forward_falls_back(golden):
    """Eligibility of the fused path also depends on the process-global diagnostic switches: asked per call, not cached."""
    from mri_super_resolution_amd._lib import lib
    net, x, d = _net(golden)
    t = dev(d["lr_pixels"][0])
    ((net(x) - t) ** 2).mean().backward()
    g_hp = [host(p.grad).copy() for p in net.parameters()]
    lib().inr_debug_set(3, 0)                       # exact-fp32 kernels only: the pre-split entry points refuse
    try:
        net.zero_grad()
        ops.launch_counts_reset()
        ((net(x) - t) ** 2).mean().backward()
        assert ops.launch_counts()["f32_pipe16"] >= 11
        for a, p in zip(g_hp, net.parameters()):
            assert O.rel_l2(host(p.grad), a) < 1e-4
    finally:
        lib().inr_debug_set(3, 1)
