"""PyTorch-CPU restatement of the RAMS multi-image network -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/multi-image-super-resolution/utils/network.py:18-155 (graph), utils/prediction.py:76-83
(predict_tensor) and master.py:43-52 (25 random 9-acquisition subsets averaged).

PARITY UNPINNED: the reference implementation is TensorFlow/Keras + tensorflow-addons; neither is installed here
and half of every shipped checkpoint is stripped, so this file cannot be checked against the
reference itself.  The TF semantics it restates -- NDHWC layout, 'same' = zero padding (odd kernels: 1 voxel each
side), REFLECT padding without edge repeat, WeightNormalization kernel = g * v / ||v|| with the norm over every
kernel axis except the output-channel one, depth_to_space in DCR order, tf.round = half-to-even -- are taken
from the libraries' documented behaviour and are pinned only by this repo's own tests.

Parameters live in a flat dict of numpy arrays keyed ``<layer>/v`` ([k1,k2(,k3),Cin,Cout] TF kernel layout),
``<layer>/g`` ([Cout]) and ``<layer>/b`` ([Cout]); ``rams_layer_names`` lists the layers in graph order.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

MEAN = 7433.6436   # network.py:18
STD = 2353.0723    # network.py:19


def rams_layer_specs(scale=3, filters=32, kernel_size=3, channels=9, r=8, N=12):
    """[(name, kernel_shape_without_channels, cin, cout)] in the order the reference builds them (network.py:119-147)."""
    k3 = (kernel_size,) * 3
    k2 = (kernel_size,) * 2
    specs = [("stem", k3, 1, filters)]

    def rfab(prefix):
        return [(f"{prefix}/conv1", k3, filters, filters), (f"{prefix}/conv2", k3, filters, filters),
                (f"{prefix}/squeeze", (1, 1, 1), filters, int(filters / r)),
                (f"{prefix}/excite", (1, 1, 1), int(filters / r), filters)]

    for i in range(N):
        specs += rfab(f"rfab{i}")
    specs.append(("trunk", k3, filters, filters))
    for i in range(channels // 3):
        specs += rfab(f"red{i}/rfab")
        specs.append((f"red{i}/conv", (3, 3, 3), filters, filters))
    specs.append(("up", (3, 3, 3), filters, scale ** 2))
    specs += [("rtab/conv1", k2, 9, 9), ("rtab/conv2", k2, 9, 9), ("rtab/squeeze", (1, 1), 9, int(9 / r)),
              ("rtab/excite", (1, 1), int(9 / r), 9), ("global", (3, 3), 9, scale ** 2)]
    return specs


def init_rams_params(seed=0, perturb_g=True, **kw):
    """Random parameters: glorot-uniform ``v`` (Keras default), small random bias, ``g`` = ||v|| (the
    WeightNormalization initial value, data_init=False) optionally perturbed so that the g/||v|| fold is exercised."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, ks, cin, cout in rams_layer_specs(**kw):
        fan_in, fan_out = int(np.prod(ks)) * cin, int(np.prod(ks)) * cout
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        v = rng.uniform(-lim, lim, size=ks + (cin, cout)).astype(np.float32)
        norm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(v.ndim - 1)))).astype(np.float32)
        g = norm * (rng.uniform(0.8, 1.25, size=cout).astype(np.float32) if perturb_g else 1.0)
        params[f"{name}/v"] = v
        params[f"{name}/g"] = g.astype(np.float32)
        params[f"{name}/b"] = rng.uniform(-0.05, 0.05, size=cout).astype(np.float32)
    return params


def effective_kernel(params, name):
    """tfa WeightNormalization: kernel = g * v / ||v||, norm over all axes but the last (network.py:29-35)."""
    v = params[f"{name}/v"].astype(np.float64)
    norm = np.sqrt((v ** 2).sum(axis=tuple(range(v.ndim - 1)), keepdims=True))
    return (params[f"{name}/g"].astype(np.float64) * v / norm).astype(np.float32)


def _conv(x, params, name, padding, dims):
    """x: [B, D1, D2(, D3), C] channels-last; TF kernel [k..., Cin, Cout]."""
    w = torch.from_numpy(effective_kernel(params, name))
    b = torch.from_numpy(params[f"{name}/b"])
    if dims == 3:
        xt = x.permute(0, 4, 1, 2, 3)
        wt = w.permute(4, 3, 0, 1, 2)
        pad = tuple(k // 2 for k in w.shape[:3]) if padding == "same" else 0
        return F.conv3d(xt, wt, b, padding=pad).permute(0, 2, 3, 4, 1)
    xt = x.permute(0, 3, 1, 2)
    wt = w.permute(3, 2, 0, 1)
    pad = tuple(k // 2 for k in w.shape[:2]) if padding == "same" else 0
    return F.conv2d(xt, wt, b, padding=pad).permute(0, 2, 3, 1)


def _reflect_hw(x):
    """tf.pad(..., [[0,0],[1,1],[1,1],...], mode='REFLECT') on axes 1 and 2 (network.py:37-39, :145)."""
    idx = lambda n: torch.tensor([1] + list(range(n)) + [n - 2])
    return x.index_select(1, idx(x.shape[1])).index_select(2, idx(x.shape[2]))


def _attention_block(x, params, prefix, dims):
    """RFAB (dims=3, network.py:42-63) / RTAB (dims=2, network.py:65-87)."""
    res = x
    y = torch.relu(_conv(x, params, f"{prefix}/conv1", "same", dims))
    y = _conv(y, params, f"{prefix}/conv2", "same", dims)
    pooled = y.mean(dim=tuple(range(1, 1 + dims)), keepdim=True)
    s = torch.relu(_conv(pooled, params, f"{prefix}/squeeze", "same", dims))
    s = torch.sigmoid(_conv(s, params, f"{prefix}/excite", "same", dims))
    return y * s + res


def depth_to_space(x, bs):
    """tf.nn.depth_to_space, NHWC, DCR: out[b, h*bs+i, w*bs+j, c] = in[b, h, w, (i*bs + j)*C + c]."""
    b, h, w, c = x.shape
    co = c // (bs * bs)
    return x.reshape(b, h, w, bs, bs, co).permute(0, 1, 3, 2, 4, 5).reshape(b, h * bs, w * bs, co)


def rams_forward(params, x, scale=3, filters=32, kernel_size=3, channels=9, r=8, N=12):
    """network.py:110-155.  x: float32 array/tensor [B, H, W, channels] -> [B, scale*H, scale*W, 1]."""
    x = torch.as_tensor(np.asarray(x, np.float32)) if not torch.is_tensor(x) else x.float()
    with torch.no_grad():
        xn = (x - MEAN) / STD
        g_res = xn
        y = _reflect_hw(xn.unsqueeze(-1))
        y = _conv(y, params, "stem", "same", 3)
        trunk_res = y
        for i in range(N):
            y = _attention_block(y, params, f"rfab{i}", 3)
        y = _conv(y, params, "trunk", "same", 3) + trunk_res
        for i in range(channels // 3):
            y = _reflect_hw(y)
            y = _attention_block(y, params, f"red{i}/rfab", 3)
            y = torch.relu(_conv(y, params, f"red{i}/conv", "valid", 3))
        y = _conv(y, params, "up", "valid", 3)[..., 0, :]
        y = depth_to_space(y, scale)
        g = _reflect_hw(g_res)
        g = _attention_block(g, params, "rtab", 2)
        g = depth_to_space(_conv(g, params, "global", "valid", 2), scale)
        return ((y + g) * STD + MEAN).numpy()


def predict_tensor(params, x, **kw):
    """prediction.py:76-83: cast -> model -> clip [0, 2**16] -> round half to even."""
    sr = rams_forward(params, np.asarray(x, np.float32), **kw)
    return np.round(np.clip(sr, 0.0, 2.0 ** 16)).astype(np.float32)


def shift_losses(y_true, y_pred, y_mask, size, border=3):
    """utils/loss.py:26-127 restated in float64: returns (min cL1 per image, max cPSNR per image)."""
    yt, yp, mk = (np.asarray(a, np.float64) for a in (y_true, y_pred, y_mask))
    c = size - 2 * border
    pred = yp[:, border:size - border, border:size - border]
    l1s, ps = [], []
    for i in range(2 * border + 1):
        for j in range(2 * border + 1):
            lab = yt[:, i:i + c, j:j + c]
            m = mk[:, i:i + c, j:j + c]
            pm, lm = pred * m, lab * m
            tot = m.sum(axis=(1, 2))
            b = ((lm - pm).sum(axis=(1, 2)) / tot)[:, None, None]
            corr = (pm + b) * m
            l1s.append(np.abs(lm - corr).sum(axis=(1, 2)) / tot)
            ps.append(10.0 * np.log10(65535.0 ** 2 / (((lm - corr) ** 2).sum(axis=(1, 2)) / tot)))
    return np.min(np.stack(l1s), axis=0), np.max(np.stack(ps), axis=0)
