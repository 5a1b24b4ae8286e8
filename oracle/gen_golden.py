#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REAL reference (build container only).

Run from the repo root:  python oracle/gen_golden.py
Needs /root/reference (absent on the GPU box -- which is why the outputs are committed).
Only data (inputs + expected outputs) is written; no reference source is copied.
"""
import hashlib
import os
import sys
import warnings

import numpy as np
import scipy.io as sio
import torch

warnings.filterwarnings("ignore")
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "implicit-neural-representations"))
import SRDWI  # noqa: E402  (the reference module itself)
import INRmodel  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def strided(a, k=97):
    """Every k-th element as an owning COPY (a 1-element slice would otherwise alias live parameter memory);
    tensors of <= 512 elements are kept whole."""
    flat = a.reshape(-1)
    return np.array(flat if flat.size <= 512 else flat[::k], copy=True)


def save(name, **kw):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def named_grads(model):
    return {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters()}


def main():
    torch.set_num_threads(8)

    # ---- grids (a-1) -------------------------------------------------------------------
    g = {}
    for n in (1, 2, 3, 5, 7, 24, 25, 28, 34, 50, 56, 60, 64, 100, 120, 128, 200, 256, 512, 1000):
        g[f"lin_{n}"] = torch.linspace(-1, 1, steps=n).numpy()
    for shape in ((64, 64), (5, 7, 3), (25, 25, 21, 4), (1, 4), (128, 128, 28)):
        m = SRDWI.get_mgrid(shape).numpy()
        tag = "x".join(map(str, shape))
        g[f"sha_{tag}"] = np.array(sha(m))
        g[f"head_{tag}"] = m[:8].copy()
        g[f"tail_{tag}"] = m[-8:].copy()
        if m.size <= 40000:
            g[f"full_{tag}"] = m
    save("grids.npz", **g)

    # ---- real input slice (cfg 1) ------------------------------------------------------
    vol = sio.loadmat(os.path.join(REF, "anon_data", "pat07_mean_b0.mat"))["data_mean_b0"]
    hr = (vol[:, :, 11] / vol[:, :, 11].max()).astype(np.float32)
    lr = np.ascontiguousarray(hr[::2, ::2])
    save("pat07_slice11.npz", hr=hr, lr=lr, vol_shape=np.array(vol.shape), vol_max=np.float32(vol.max()))
    save("pat07_volume.npz", vol=np.ascontiguousarray(vol.astype(np.float32)))   # config 2 input (whole volume)

    # ---- dataset flattening (a-2) + Fourier features (a-3) -----------------------------
    ds = SRDWI.ImageFitting_set([lr.astype(np.float64)])
    rng = np.random.default_rng(5)
    img3 = rng.random((5, 7, 3))
    ds3 = SRDWI.ImageFitting_set([img3, img3 * 2])
    Bs = {}
    for d in (2, 3, 4):
        np.random.seed(0)
        Bs[d] = (np.random.normal(size=(128, d)) * 0.5).astype(np.float32)
    B2 = torch.from_numpy(Bs[2])
    model_input = SRDWI.input_mapping(ds.coords[0], B2)
    x3 = SRDWI.get_mgrid((5, 7, 3))
    x4 = SRDWI.get_mgrid((3, 4, 2, 4))
    save("dataset_ff.npz",
         lr_pixels=ds.pixels.numpy(), lr_coords_sha=np.array(sha(ds.coords.numpy())),
         img3=img3, ds3_pixels=ds3.pixels.numpy(), ds3_coords=ds3.coords.numpy(),
         B2=Bs[2], B3=Bs[3], B4=Bs[4],
         ff2_rows=model_input.numpy()[::17].copy(), ff2_sha=np.array(sha(model_input.numpy())),
         ff3=SRDWI.input_mapping(x3, torch.from_numpy(Bs[3])).numpy(),
         ff4=SRDWI.input_mapping(x4, torch.from_numpy(Bs[4])).numpy())

    # ---- Siren(256,512,3,1): init, forward, grads, short Adam trajectories -------------
    target = ds.pixels[0]
    hr_grid_in = SRDWI.input_mapping(SRDWI.get_mgrid((128, 128)), B2)
    out = {}
    for flavor, mod in (("SRDWI", SRDWI), ("INRmodel", INRmodel)):
        torch.manual_seed(0)
        net = mod.Siren(256, 512, 3, 1)
        for n, p in net.named_parameters():
            a = p.detach().numpy()
            out[f"{flavor}/init_sha/{n}"] = np.array(sha(a))
            out[f"{flavor}/init_sum/{n}"] = np.float64(a.astype(np.float64).sum())
        y = net(model_input)
        out[f"{flavor}/fwd"] = y.detach().numpy().copy()
        loss = ((y - target) ** 2).mean()
        loss.backward()
        out[f"{flavor}/loss0"] = np.float32(loss.item())
        for n, gr in named_grads(net).items():
            out[f"{flavor}/grad_norm/{n}"] = np.float64(np.linalg.norm(gr.astype(np.float64)))
            out[f"{flavor}/grad_strided/{n}"] = strided(gr)
            if gr.size <= 512:
                out[f"{flavor}/grad_full/{n}"] = gr
    save("siren512_step0.npz", **out)

    traj = {}
    for threads in (8, 1):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        net = SRDWI.Siren(256, 512, 3, 1)
        opt = torch.optim.Adam(lr=1e-4, params=list(net.parameters()))
        losses = []
        for step in range(1, 51):
            y = net(model_input)
            loss = ((y - target) ** 2).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
            if step in (1, 10, 50):
                with torch.no_grad():
                    rec = torch.clamp(net(hr_grid_in), min=0).view(128, 128).numpy().copy()
                traj[f"t{threads}/recon_{step}"] = rec
                for n, p in net.named_parameters():
                    a = p.detach().numpy()
                    traj[f"t{threads}/pnorm_{step}/{n}"] = np.float64(np.linalg.norm(a.astype(np.float64)))
                    traj[f"t{threads}/pstrided_{step}/{n}"] = strided(a, 997)
        traj[f"t{threads}/losses"] = np.array(losses, np.float64)
    torch.set_num_threads(8)
    save("siren512_traj.npz", **traj)

    # ---- master.py-style small 2-D net: Siren(2,64,6,1), raw coords, weighted loss ------
    roi = hr[40:100, 40:100]
    coords2 = SRDWI.get_mgrid((60, 60))
    tgt2 = torch.from_numpy((2.0 * roi - 1.0).reshape(-1, 1).copy())  # Normalize(0.5,0.5): nn_mri.py:174-180
    wts = torch.from_numpy((np.random.default_rng(3).random((3600, 1)) > 0.2).astype(np.float32))
    small = {"coords": coords2.numpy(), "target": tgt2.numpy(), "weight": wts.numpy()}
    torch.manual_seed(0)
    net = SRDWI.Siren(2, 64, 6, 1)
    for n, p in net.named_parameters():
        small[f"init/{n}"] = p.detach().numpy().copy()
    y = net(coords2)
    small["fwd"] = y.detach().numpy().copy()
    loss = (wts * (y - tgt2) ** 2).mean()
    loss.backward()
    small["loss0"] = np.float32(loss.item())
    for n, gr in named_grads(net).items():
        small[f"grad/{n}"] = gr
    opt = torch.optim.Adam(lr=3e-4, params=net.parameters())
    big_grid = SRDWI.get_mgrid((180, 180))
    losses = []
    for step in range(1, 51):
        y = net(coords2)
        loss = (wts * (y - tgt2) ** 2).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if step in (1, 10, 50):
            for n, p in net.named_parameters():
                small[f"p{step}/{n}"] = p.detach().numpy().copy()
            with torch.no_grad():
                small[f"recon180_{step}"] = net(big_grid).view(180, 180).numpy().copy()
    small["losses"] = np.array(losses, np.float64)
    save("siren64_2d.npz", **small)

    # ---- shipped checkpoint model.pt: forward on a 128x128 grid -------------------------
    sd = torch.load(os.path.join(REF, "implicit-neural-representations", "model.pt"),
                    weights_only=True, map_location="cpu")
    net = SRDWI.Siren(2, 64, 3, 1)
    net.load_state_dict({k: v for k, v in sd.items() if k.startswith("net.")}, strict=False)
    mp = {k.replace(".", "__"): v.numpy() for k, v in sd.items() if k.startswith("net.")}
    with torch.no_grad():
        mp["fwd128"] = net(SRDWI.get_mgrid((128, 128))).view(128, 128).numpy().copy()
    save("model_pt.npz", **mp)

    # ---- PerturbNet forward (SRDWI.PN hard-codes .cuda(): run it on CPU via an identity shim)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.manual_seed(0)
    pn = SRDWI.PN(256, 128, 3)
    rows = SRDWI.input_mapping(x3, torch.from_numpy(Bs[3]))
    pnd = {f"param/{n}": p.detach().numpy().copy() for n, p in pn.named_parameters()}
    pnd["in"] = rows.numpy()
    with torch.no_grad():
        pnd["out_s3"] = pn(rows, 3, 1 / 128.).numpy().copy()
        pnd["out_s0"] = pn(rows, 0, 1 / 128.).numpy().copy()
    save("pn.npz", **pnd)

    # ---- host-side helpers: ADC, z-resize ----------------------------------------------
    rng = np.random.default_rng(11)
    bvals = np.array([0.0, 150.0, 1000.0, 1500.0])
    sl = rng.random((6, 5, 4)) + 0.05
    save("helpers.npz", bvals=bvals, slicedata=sl, adc=SRDWI.calculate_ADC(bvals, sl),
         resize_in=sl, resize_out=SRDWI.resize_array(sl, 9))


if __name__ == "__main__":
    main()
