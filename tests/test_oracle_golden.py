"""Pins the CPU oracle (oracle/inr_oracle.py, oracle/torch_port.py) against fixtures produced by
importing the real reference (oracle/gen_golden.py).  CPU only."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import inr_oracle as O
from oracle import torch_port as P
from conftest import strided_sample


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


PARAM_ORDER_SRDWI = ["final_linear.weight", "final_linear.bias"] + [
    f"net.{l}.linear.{k}" for l in range(4) for k in ("weight", "bias")]


def test_linspace_bit_exact(golden):
    g = golden("grids.npz")
    for key in g.files:
        if key.startswith("lin_"):
            n = int(key[4:])
            assert np.array_equal(bits(O.linspace_pm1(n)), bits(g[key])), n


@pytest.mark.parametrize("shape", [(64, 64), (5, 7, 3), (25, 25, 21, 4), (1, 4), (128, 128, 28)])
def test_mgrid_bit_exact(golden, shape):
    g = golden("grids.npz")
    tag = "x".join(map(str, shape))
    m = O.mgrid(shape)
    assert m.shape == (int(np.prod(shape)), len(shape))
    assert sha(m) == str(g[f"sha_{tag}"])
    assert np.array_equal(bits(m[:8]), bits(g[f"head_{tag}"]))
    assert np.array_equal(bits(m[-8:]), bits(g[f"tail_{tag}"]))
    assert np.array_equal(bits(P.port_mgrid(shape).numpy()), bits(m))


def test_dataset_flattening(golden):
    g = golden("dataset_ff.npz")
    lr = golden("pat07_slice11.npz")["lr"]
    pixels, coords, shape = O.image_fitting_set([lr.astype(np.float64)])
    assert shape == (64, 64)
    assert np.array_equal(bits(pixels), bits(g["lr_pixels"]))
    assert sha(coords) == str(g["lr_coords_sha"])
    img3 = g["img3"]
    p3, c3, _ = O.image_fitting_set([img3, img3 * 2])
    assert np.array_equal(bits(p3), bits(g["ds3_pixels"]))
    assert np.array_equal(bits(c3), bits(g["ds3_coords"]))


def test_fourier_features(golden):
    g = golden("dataset_ff.npz")
    ff2 = O.fourier_features(O.mgrid((64, 64)), g["B2"])
    assert ff2.shape == (4096, 256)
    assert O.rel_l2(ff2[::17], g["ff2_rows"]) < 2e-6
    assert O.rel_l2(O.fourier_features(O.mgrid((5, 7, 3)), g["B3"]), g["ff3"]) < 2e-6
    assert O.rel_l2(O.fourier_features(O.mgrid((3, 4, 2, 4)), g["B4"]), g["ff4"]) < 2e-6
    # identity when B is None (SRDWI.py:112-113)
    x = O.mgrid((3, 3))
    assert O.fourier_features(x, None) is x
    # torch port is the same op sequence -> bit-identical on this machine
    t = P.port_input_mapping(P.port_mgrid((5, 7, 3)), torch.from_numpy(g["B3"])).numpy()
    assert O.rel_l2(t, g["ff3"]) < 1e-7


def test_fourier_matrix_seed(golden):
    g = golden("dataset_ff.npz")
    for d in (2, 3, 4):
        assert np.array_equal(bits(P.fourier_matrix(d)), bits(g[f"B{d}"]))


@pytest.mark.parametrize("flavor", ["SRDWI", "INRmodel"])
def test_port_init_matches_reference_rng_order(golden, flavor):
    g = golden("siren512_step0.npz")
    torch.manual_seed(0)
    net = P.PortSiren(256, 512, 3, 1, flavor=flavor)
    names = [n for n, _ in net.named_parameters()]
    ref_names = [k.split("/", 2)[2] for k in g.files if k.startswith(f"{flavor}/init_sha/")]
    assert names == ref_names
    for n, p in net.named_parameters():
        assert sha(p.detach().numpy()) == str(g[f"{flavor}/init_sha/{n}"]), n
    assert set(net.state_dict().keys()) == set(names) | {"net.4.weight", "net.4.bias"}


def _step0(golden, flavor):
    g = golden("siren512_step0.npz")
    d = golden("dataset_ff.npz")
    torch.manual_seed(0)
    net = P.PortSiren(256, 512, 3, 1, flavor=flavor)
    x = O.fourier_features(O.mgrid((64, 64)), d["B2"])
    return g, d, net, x


@pytest.mark.parametrize("flavor", ["SRDWI", "INRmodel"])
def test_forward_numpy_and_port(golden, flavor):
    g, d, net, x = _step0(golden, flavor)
    ws, bs = net.layer_params()
    y32 = O.siren_forward(ws, bs, x)
    y64 = O.siren_forward(ws, bs, x.astype(np.float64), dtype=np.float64)
    ref = g[f"{flavor}/fwd"]
    assert O.rel_l2(y32, ref) < 5e-6
    assert O.rel_l2(y64, ref) < 2e-6
    yt = net(torch.from_numpy(x)).detach().numpy()
    assert O.rel_l2(yt, ref) < 2e-6


def test_loss_and_gradients(golden):
    g, d, net, x = _step0(golden, "SRDWI")
    ws, bs = net.layer_params()
    t = d["lr_pixels"][0]
    y, acts, pre = O.siren_forward([w.astype(np.float64) for w in ws], [b.astype(np.float64) for b in bs],
                                   x.astype(np.float64), dtype=np.float64, stash=True)
    loss, gy = O.mse_loss_and_grad(y, t, dtype=np.float64)
    assert abs(loss - g["SRDWI/loss0"]) / g["SRDWI/loss0"] < 1e-5
    gw, gb = O.siren_backward(ws, acts, pre, gy, dtype=np.float64)
    # network order -> reference parameter names
    byname = {"final_linear.weight": gw[-1], "final_linear.bias": gb[-1]}
    for l in range(4):
        byname[f"net.{l}.linear.weight"] = gw[l]
        byname[f"net.{l}.linear.bias"] = gb[l]
    for n in PARAM_ORDER_SRDWI:
        ref_s = g[f"SRDWI/grad_strided/{n}"]
        assert O.rel_l2(strided_sample(byname[n]), ref_s) < 1e-5, n
        nrm = np.linalg.norm(byname[n])
        assert abs(nrm - g[f"SRDWI/grad_norm/{n}"]) / g[f"SRDWI/grad_norm/{n}"] < 1e-5, n
    # torch port autograd: same ops as the reference
    out = net(torch.from_numpy(x))
    ((out - torch.from_numpy(t)) ** 2).mean().backward()
    for n, p in net.named_parameters():
        assert O.rel_l2(strided_sample(p.grad.numpy()), g[f"SRDWI/grad_strided/{n}"]) < 1e-5, n


def test_short_trajectory_port_and_numpy_adam(golden):
    """Port fit == reference fit for 10 steps (T3); numpy Adam == torch Adam on the same grads."""
    tr = golden("siren512_traj.npz")
    g, d, net, x = _step0(golden, "SRDWI")
    xt, tt = torch.from_numpy(x), torch.from_numpy(d["lr_pixels"][0])
    # numpy Adam shadow of the head bias + one weight tensor
    shadow = {n: (p.detach().numpy().copy(), np.zeros(p.shape, np.float32), np.zeros(p.shape, np.float32))
              for n, p in net.named_parameters() if n in ("final_linear.bias", "net.3.linear.bias")}
    step_box = [0]

    opt = torch.optim.Adam(lr=1e-4, params=list(net.parameters()))

    def hook(k, model):
        step_box[0] = k
        for n, p in model.named_parameters():
            if n in shadow:
                pp, m, v = shadow[n]
                O.adam_step(pp, p.grad.numpy(), m, v, k, 1e-4)
                assert np.allclose(pp, p.detach().numpy(), rtol=2e-6, atol=1e-9), (n, k)

    losses, _ = P.port_fit(net, xt, tt, 10, optimizer=opt, on_step=hook)
    assert np.allclose(losses, tr["t8/losses"][:10], rtol=1e-4)
    hr_in = O.fourier_features(O.mgrid((128, 128)), d["B2"])
    with torch.no_grad():
        rec = torch.clamp(net(torch.from_numpy(hr_in)), min=0).view(128, 128).numpy()
    assert O.rel_l2(rec, tr["t8/recon_10"]) < 1e-4
    # the reference's own thread-count noise at step 10 / 50 documents the tolerance
    assert O.rel_l2(tr["t1/recon_10"], tr["t8/recon_10"]) < 1e-4
    assert O.rel_l2(tr["t1/recon_50"], tr["t8/recon_50"]) < 1e-4


def test_small_2d_net_weighted(golden):
    s = golden("siren64_2d.npz")
    torch.manual_seed(0)
    net = P.PortSiren(2, 64, 6, 1)
    for n, p in net.named_parameters():
        assert np.array_equal(bits(p.detach().numpy()), bits(s[f"init/{n}"])), n
    ws, bs = net.layer_params()
    y, acts, pre = O.siren_forward(ws, bs, s["coords"].astype(np.float64), dtype=np.float64, stash=True)
    assert O.rel_l2(y, s["fwd"]) < 2e-5
    loss, gy = O.mse_loss_and_grad(y, s["target"], s["weight"], dtype=np.float64)
    assert abs(loss - s["loss0"]) / s["loss0"] < 1e-5
    gw, gb = O.siren_backward(ws, acts, pre, gy, dtype=np.float64)
    assert O.rel_l2(gw[-1], s["grad/final_linear.weight"]) < 5e-5
    for l in range(7):
        assert O.rel_l2(gw[l], s[f"grad/net.{l}.linear.weight"]) < 5e-5, l
        assert O.rel_l2(gb[l], s[f"grad/net.{l}.linear.bias"]) < 5e-5, l
    losses, _ = P.port_fit(net, torch.from_numpy(s["coords"]), torch.from_numpy(s["target"]), 10,
                           lr=3e-4, weight=torch.from_numpy(s["weight"]))
    assert np.allclose(losses, s["losses"][:10], rtol=1e-4)
    for n, p in net.named_parameters():
        assert O.rel_l2(p.detach().numpy(), s[f"p10/{n}"]) < 1e-4, n


def test_model_pt_forward(golden):
    m = golden("model_pt.npz")
    ws = [m[f"net__{l}__linear__weight"] for l in range(4)] + [m["net__4__weight"]]
    bs = [m[f"net__{l}__linear__bias"] for l in range(4)] + [m["net__4__bias"]]
    y = O.siren_forward(ws, bs, O.mgrid((128, 128)).astype(np.float64), dtype=np.float64)
    assert O.rel_l2(y.reshape(128, 128), m["fwd128"]) < 2e-5
    rec = O.reconstruct(ws, bs, (128, 128), None, clamp_min=None, dtype=np.float64)
    assert O.rel_l2(rec, m["fwd128"]) < 2e-5


def test_pn_forward(golden):
    p = golden("pn.npz")
    args = (p["param/perturb_linear.weight"], p["param/perturb_linear.bias"],
            p["param/perturb_linear2.weight"], p["param/perturb_linear2.bias"], p["in"])
    assert O.rel_l2(O.pn_forward(*args, sample=3, eps=1 / 128.), p["out_s3"]) < 5e-6
    assert O.rel_l2(O.pn_forward(*args, sample=0, eps=1 / 128.), p["out_s0"]) < 5e-6
    torch.manual_seed(0)
    pn = P.PortPN(256, 128, 3)
    for n, q in pn.named_parameters():
        assert np.array_equal(bits(q.detach().numpy()), bits(p[f"param/{n}"]))
    with torch.no_grad():
        assert O.rel_l2(pn(torch.from_numpy(p["in"]), 3, 1 / 128.).numpy(), p["out_s3"]) < 1e-6


def test_adc_closed_form(golden):
    h = golden("helpers.npz")
    assert np.allclose(O.adc_map(h["bvals"], h["slicedata"]), h["adc"], rtol=1e-9, atol=1e-12)


def test_psnr_ssim_analytic():
    rng = np.random.default_rng(0)
    a = rng.random((40, 40))
    assert O.ssim2d(a, a) == pytest.approx(1.0, abs=1e-12)
    assert O.psnr(a, a + 0.1) == pytest.approx(20.0, abs=1e-9)
    # constant images: ssim = (2ab+c1)/(a^2+b^2+c1)
    x, y = np.full((20, 20), 0.3), np.full((20, 20), 0.5)
    c1 = 0.01 ** 2
    assert O.ssim2d(x, y) == pytest.approx((2 * .15 + c1) / (.09 + .25 + c1), rel=1e-9)


# ---- (f)-1 three-compartment hybrid fit: restated scipy TRF pinned against the reference's hybrid_fit ----------------
def test_hybrid_fit_oracle_vs_reference(golden):
    from oracle import pia_oracle as PIA
    from tests import pia_common as C
    g = golden("pia_hybrid.npz")
    sel = np.arange(0, 128, 2)      # 64 voxels, all four noise levels
    x, cost, nfev = [], [], []
    for y in g["signals"][sel]:
        p, info = PIA.trf_fit(y, return_info=True)
        x.append(p); cost.append(info["cost"]); nfev.append(info["nfev"])
    ref = C.pack(g["D"], g["T2"], g["v"])[sel]
    C.check_against(np.array(x), ref, np.array(cost), g["cost"][sel], np.array(nfev), g["nfev"][sel])
    D, T2, v = PIA.hybrid_fit(g["signals"][:3])
    assert np.allclose(v.sum(axis=1), 1.0) and D.shape == T2.shape == v.shape == (3, 3)


def test_hybrid_fit_pieces_match_scipy():
    """The restated building blocks against scipy's own (the reference's dependency), on random states incl. bounds."""
    from scipy.optimize._lsq import common as SC, trf as ST
    from scipy.optimize._numdiff import approx_derivative
    from oracle import pia_oracle as PIA
    rng = np.random.default_rng(0)
    for trial in range(150):
        y = PIA.synthetic_signals(1, 0.05, seed=trial)[0]
        fun = lambda p: PIA.three_compartment(p) - y
        x = rng.uniform(PIA.LB, PIA.UB)
        k = rng.integers(0, 8, size=2)
        if trial % 3 == 0:
            x[k[0]] = np.nextafter(PIA.LB[k[0]], PIA.UB[k[0]])
        if trial % 3 == 1:
            x[k[1]] = np.nextafter(PIA.UB[k[1]], PIA.LB[k[1]])
        f = fun(x)
        J = PIA._fd_jacobian(fun, x, f)
        assert np.array_equal(J, approx_derivative(fun, x, f0=f, bounds=(PIA.LB, PIA.UB), method="2-point"))
        g = J.T @ f
        v, dv = PIA._cl_scaling(x, g)
        v2, dv2 = SC.CL_scaling_vector(x, g, PIA.LB, PIA.UB)
        assert np.array_equal(v, v2) and np.array_equal(dv, dv2)
        d, diag, gh, Jh = v ** 0.5, g * dv, v ** 0.5 * g, J * v ** 0.5
        U, s, Vt = np.linalg.svd(np.vstack([Jh, np.diag(diag ** 0.5)]), full_matrices=False)
        uf = U.T @ np.concatenate([f, np.zeros(8)])
        delta = 10 ** rng.uniform(-3, 3)
        a0 = 0.0 if trial % 2 else 10 ** rng.uniform(-3, 3)
        p1, al1 = PIA._solve_tr(16, U[:16].T @ f, s, Vt.T, delta, a0)
        p2, al2, _ = SC.solve_lsq_trust_region(8, 16, uf, s, Vt.T, delta, initial_alpha=a0)
        assert np.allclose(p1, p2, rtol=1e-12, atol=0) and al1 == pytest.approx(al2, rel=1e-12)
        theta = max(0.995, 1 - np.abs(g * v).max())
        s1 = PIA._select_step(x, Jh, diag, gh, d * p1, p1.copy(), d, delta, theta)
        s2 = ST.select_step(x, Jh, diag, gh, d * p2, p2.copy(), d, delta, PIA.LB, PIA.UB, theta)
        for a, b in zip(s1, s2):
            assert np.allclose(a, b, rtol=1e-10, atol=1e-300)
        xb = x + s1[0]
        xb[k[0]], xb[k[1]] = PIA.LB[k[0]], PIA.UB[k[1]]
        assert np.array_equal(PIA._strictly_feasible(xb), SC.make_strictly_feasible(xb, PIA.LB, PIA.UB, rstep=0))
