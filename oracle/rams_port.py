"""PyTorch-CPU restatement of the RAMS multi-image network -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/multi-image-super-resolution/utils/network.py:18-155 (graph), utils/prediction.py:76-83
(predict_tensor) and master.py:43-52 (25 random 9-acquisition subsets averaged).

PARITY UNPINNED: the reference implementation is TensorFlow/Keras + tensorflow-addons; neither is installed here
and half of every shipped checkpoint is stripped, so this file cannot be checked against the
reference itself.  The TF semantics it restates -- NDHWC layout, 'same' = zero padding (odd kernels: 1 voxel each
side), REFLECT padding without edge repeat, WeightNormalization kernel = g * v / ||v|| with the norm over every
kernel axis except the output-channel one, depth_to_space in DCR order, tf.round = half-to-even -- are taken
from the libraries' documented behaviour and are pinned only by this repo's own tests.

Parameters live in a flat dict of numpy arrays (or torch tensors, for autograd) keyed ``<layer>/v`` ([k1,k2(,k3),Cin,Cout]
TF kernel layout), ``<layer>/g`` ([Cout]) and ``<layer>/b`` ([Cout]).  The layer list is NOT a table: ``rams_layer_specs``
runs the graph once with a recording parameter store, so it is derived from this file's forward pass alone and is
independent of the product's ``rams.rams_layer_specs`` (tests compare the two).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

MEAN = 7433.6436   # network.py:18
STD = 2353.0723    # network.py:19


class _Recorder(dict):
    """Parameter store that records (name, kernel shape, cin, cout) of every convolution the forward pass BUILDS, in
    build order -- the way the Keras functional graph of network.py:110-155 comes into being.  Missing entries are
    created on first use (zeros), so one dry forward on a tiny input yields the layer list without any table."""

    def __init__(self):
        super().__init__()
        self.order = []

    def declare(self, name, ks, cin, cout):
        if f"{name}/v" not in self:
            self.order.append((name, tuple(ks), int(cin), int(cout)))
            self[f"{name}/v"] = np.zeros(tuple(ks) + (cin, cout), np.float32)
            self[f"{name}/g"] = np.ones(cout, np.float32)
            self[f"{name}/b"] = np.zeros(cout, np.float32)


def _kernel(params, name, dtype):
    """tfa WeightNormalization: kernel = g * v / ||v||, norm over all axes but the last (network.py:29-35)."""
    v, g = params[f"{name}/v"], params[f"{name}/g"]
    if not torch.is_tensor(v):
        v, g = torch.from_numpy(np.asarray(v)), torch.from_numpy(np.asarray(g))
    v, g = v.to(torch.float64), g.to(torch.float64)
    norm = torch.sqrt((v ** 2).sum(dim=tuple(range(v.dim() - 1)), keepdim=True))
    w = g * v / norm
    # the forward pass of the fp32 pipeline uses the fp32-rounded folded kernel; autograd (float64 runs) sees the fold
    return w.to(dtype)


def effective_kernel(params, name):
    return _kernel(params, name, torch.float32).detach().numpy()


def _conv(x, params, name, cout, ks, padding):
    """x: [B, D1, D2(, D3), C] channels-last; TF kernel [k..., Cin, Cout]; `cout` / `ks` as the reference declares the
    layer (conv3d_weightnorm(filters, kernel_size, ...), network.py:29-35)."""
    dims = len(ks)
    if isinstance(params, _Recorder):
        params.declare(name, ks, x.shape[-1], cout)
    w = _kernel(params, name, x.dtype)
    b = params[f"{name}/b"]
    b = (b if torch.is_tensor(b) else torch.from_numpy(np.asarray(b))).to(x.dtype)
    if dims == 3:
        xt = x.permute(0, 4, 1, 2, 3)
        wt = w.permute(4, 3, 0, 1, 2)
        pad = tuple(k // 2 for k in ks) if padding == "same" else 0
        return F.conv3d(xt, wt, b, padding=pad).permute(0, 2, 3, 4, 1)
    xt = x.permute(0, 3, 1, 2)
    wt = w.permute(3, 2, 0, 1)
    pad = tuple(k // 2 for k in ks) if padding == "same" else 0
    return F.conv2d(xt, wt, b, padding=pad).permute(0, 2, 3, 1)


def _reflect_hw(x):
    """tf.pad(..., [[0,0],[1,1],[1,1],...], mode='REFLECT') on axes 1 and 2 (network.py:37-39, :145)."""
    idx = lambda n: torch.tensor([1] + list(range(n)) + [n - 2])
    return x.index_select(1, idx(x.shape[1])).index_select(2, idx(x.shape[2]))


def _attention_block(x, params, prefix, filters, ks, r):
    """RFAB (3-D kernel, network.py:42-63) / RTAB (2-D kernel, network.py:65-87)."""
    dims = len(ks)
    one = (1,) * dims
    res = x
    y = torch.relu(_conv(x, params, f"{prefix}/conv1", filters, ks, "same"))
    y = _conv(y, params, f"{prefix}/conv2", filters, ks, "same")
    pooled = y.mean(dim=tuple(range(1, 1 + dims)), keepdim=True)
    s = torch.relu(_conv(pooled, params, f"{prefix}/squeeze", int(filters / r), one, "same"))
    s = torch.sigmoid(_conv(s, params, f"{prefix}/excite", filters, one, "same"))
    return y * s + res


def depth_to_space(x, bs):
    """tf.nn.depth_to_space, NHWC, DCR: out[b, h*bs+i, w*bs+j, c] = in[b, h, w, (i*bs + j)*C + c]."""
    b, h, w, c = x.shape
    co = c // (bs * bs)
    return x.reshape(b, h, w, bs, bs, co).permute(0, 1, 3, 2, 4, 5).reshape(b, h * bs, w * bs, co)


def rams_graph(params, x, scale=3, filters=32, kernel_size=3, channels=9, r=8, N=12, shuffle=None):
    """network.py:110-155 on a torch tensor x [B, H, W, channels] (any float dtype; differentiable).

    ``shuffle``: block size of the two ``depth_to_space`` calls.  The reference writes the literal 3 (network.py:141,148) beside
    ``Conv3D(scale**2)``, so with the reference's own semantics (``shuffle=None`` -> 3) any ``scale != 3`` is a TensorFlow shape
    error (4 or 16 channels are not a multiple of 9).  ``shuffle=scale`` is the build-side reading "the 3 means scale" that
    BASELINE config 3's "x4" needs; tests of scale 2 / 4 pass it explicitly."""
    bs = 3 if shuffle is None else int(shuffle)
    k3, k2 = (kernel_size,) * 3, (kernel_size,) * 2
    xn = (x - MEAN) / STD                                                          # normalize, :21-23
    g_res = xn
    y = _reflect_hw(xn.unsqueeze(-1))                                              # :117-119
    y = _conv(y, params, "stem", filters, k3, "same")                              # low-level features, :121
    trunk_res = y
    for i in range(N):                                                             # residual feature attention blocks
        y = _attention_block(y, params, f"rfab{i}", filters, k3, r)
    y = _conv(y, params, "trunk", filters, k3, "same") + trunk_res                 # :128-129
    for i in range(int(np.floor_divide(channels, 3))):                             # temporal reduction, :132-137
        y = _reflect_hw(y)
        y = _attention_block(y, params, f"red{i}/rfab", filters, k3, r)
        y = torch.relu(_conv(y, params, f"red{i}/conv", filters, (3, 3, 3), "valid"))
    y = _conv(y, params, "up", scale ** 2, (3, 3, 3), "valid")[..., 0, :]            # upscaling, :139-141
    y = depth_to_space(y, bs)                                                      # the reference hard-codes 3, :141
    g = _reflect_hw(g_res)                                                         # global path, :144-148
    g = _attention_block(g, params, "rtab", 9, k2, r)                              # RTAB(x, 9, ...), :146
    g = depth_to_space(_conv(g, params, "global", scale ** 2, (3, 3), "valid"), bs)
    return (y + g) * STD + MEAN                                                    # denormalize, :25-27


def rams_layer_specs(scale=3, filters=32, kernel_size=3, channels=9, r=8, N=12, shuffle=None):
    """[(name, kernel shape, cin, cout)] discovered by one dry run of ``rams_graph`` on an 8 x 8 input (no table)."""
    rec = _Recorder()
    with torch.no_grad():
        rams_graph(rec, torch.zeros(1, 8, 8, channels), scale, filters, kernel_size, channels, r, N, shuffle)
    return rec.order


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


def rams_forward(params, x, **kw):
    """network.py:110-155.  x: float32 array/tensor [B, H, W, channels] -> numpy [B, scale*H, scale*W, 1]."""
    x = torch.as_tensor(np.asarray(x, np.float32)) if not torch.is_tensor(x) else x.float()
    with torch.no_grad():
        return rams_graph(params, x, **kw).numpy()


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


def shift_l1_torch(y_true, y_pred, y_mask, size, border=3):
    """utils/loss.py:26-75 on torch tensors [B, size, size] (differentiable; float64 recommended): per-image minimum over
    the 7 x 7 label shifts of the brightness-corrected masked L1."""
    c = size - 2 * border
    pred = y_pred[:, border:size - border, border:size - border]
    vals = []
    for i in range(2 * border + 1):
        for j in range(2 * border + 1):
            lab = y_true[:, i:i + c, j:j + c]
            m = y_mask[:, i:i + c, j:j + c]
            pm, lm = pred * m, lab * m
            tot = m.sum(dim=(1, 2))
            b = ((lm - pm).sum(dim=(1, 2)) / tot)[:, None, None]
            corr = (pm + b) * m
            vals.append((lm - corr).abs().sum(dim=(1, 2)) / tot)
    return torch.stack(vals).min(dim=0).values


def train_grads(params, x, y_true, y_mask, dtype=torch.float64, **kw):
    """Trainer.train_step's gradient (utils/training.py:193-209) by autograd on this restatement: returns (per-image loss,
    {name: d sum(loss) / d variable})."""
    tp = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in params.items()}
    xt = torch.as_tensor(np.asarray(x), dtype=dtype)
    sr = rams_graph(tp, xt, **kw)[..., 0]
    size = sr.shape[1]
    loss = shift_l1_torch(torch.as_tensor(np.asarray(y_true), dtype=dtype), sr, torch.as_tensor(np.asarray(y_mask), dtype=dtype), size)
    loss.sum().backward()
    return loss.detach().numpy(), {k: v.grad.numpy() for k, v in tp.items()}


def keras_adam_step(params, grads, m, v, t, lr=5e-4, b1=0.9, b2=0.999, eps=1e-7):
    """Keras Adam (non-amsgrad), in place on float64 dicts: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); p -= lr_t m / (sqrt(v) + eps)."""
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    for k in params:
        m[k] = b1 * m[k] + (1 - b1) * grads[k]
        v[k] = b2 * v[k] + (1 - b2) * grads[k] ** 2
        params[k] = params[k] - lr_t * m[k] / (np.sqrt(v[k]) + eps)


# ---- RAMS+ ensembling (utils/prediction.py:10-74) restated on numpy: TEST INFRASTRUCTURE like the rest of this file -------------
def np_flip(X, rn):
    """prediction.py:54-58 (tf.image.flip_left_right = the W axis, -2)."""
    return (X if rn <= 0.5 else np.flip(X, axis=-2)), np.rint(rn)


def np_rotate(X, k):
    """prediction.py:61-65 (tf.image.rot90: counter-clockwise quarter turns of (H, W) = axes (-3, -2))."""
    return np.rot90(X, int(k) % 4, axes=(-3, -2)), k


def np_geometric_ensemble(X):
    """prediction.py:32-42 without the shuffle."""
    r = np.array(np.meshgrid([0, 1], [0, 1, 2, 3])).T.reshape(-1, 2)
    return np.stack([np_rotate(np_flip(X, r[i, 0])[0], r[i, 1])[0] for i in range(8)]), r


def np_unensemble(X, r):
    """prediction.py:45-51."""
    return np.mean([np_flip(np_rotate(X[i], 4 - r[i, 1])[0], r[i, 0])[0] for i in range(len(X))], axis=0, keepdims=True)
