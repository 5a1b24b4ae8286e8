"""RAMS multi-image super-resolution (forward / inference) on the HIP kernels.

Mirrors the reference's surface: ``RAMS(scale, filters, kernel_size, channels, r, N)``
(multi-image-super-resolution/utils/network.py:91-155) returns a callable model, ``predict_tensor(model, x)`` is
utils/prediction.py:76-83, ``predict_case`` is the 25-random-subsets loop of
multi-image-super-resolution/master.py:43-52 (run as ONE batched forward).

Parameters are kept in the reference's form -- per layer the WeightNormalization variables ``v`` (TensorFlow kernel
layout ``[k..., Cin, Cout]``), ``g`` (``[Cout]``) and the bias ``b`` -- in ``model.params`` (numpy, name -> array;
layer names as in ``rams_layer_names``).  ``model.pack()`` folds ``g*v/||v||`` and lays everything out as the one
flat device buffer ``inr_rams_forward`` consumes (layout: include/inrhip.h).  No TensorFlow checkpoint reader is
provided: the shipped checkpoints are incomplete.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import ops
from ._lib import RamsDesc, check, lib

MEAN = 7433.6436   # network.py:18
STD = 2353.0723    # network.py:19


def rams_layer_specs(scale=3, filters=32, kernel_size=3, channels=9, r=8, N=12):
    """[(name, kernel_spatial_shape, cin, cout)] in graph order (network.py:119-147)."""
    k3, k2 = (kernel_size,) * 3, (kernel_size,) * 2
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


class RAMS:
    """``RAMS(scale, filters, kernel_size, channels, r, N)``; call it on ``[B, H, W, channels]`` data."""

    def __init__(self, scale=3, filters=32, kernel_size=3, channels=9, r=8, N=12, params=None, seed=None):
        self.cfg = dict(scale=scale, filters=filters, kernel_size=kernel_size, channels=channels, r=r, N=N)
        self.desc = RamsDesc(scale, filters, kernel_size, channels, r, N, MEAN, STD)
        self.specs = rams_layer_specs(**self.cfg)
        if params is None:   # Keras defaults: glorot-uniform kernels, zero bias; WeightNormalization g = ||v||
            rng = np.random.default_rng(seed)
            params = {}
            for name, ks, cin, cout in self.specs:
                fan_in, fan_out = int(np.prod(ks)) * cin, int(np.prod(ks)) * cout
                lim = np.sqrt(6.0 / (fan_in + fan_out))
                v = rng.uniform(-lim, lim, size=ks + (cin, cout)).astype(np.float32)
                params[f"{name}/v"] = v
                params[f"{name}/g"] = np.sqrt((v.astype(np.float64) ** 2).sum(axis=tuple(range(v.ndim - 1)))).astype(np.float32)
                params[f"{name}/b"] = np.zeros(cout, np.float32)
        self.params = params
        self._packed = None

    # ---- parameter packing ---------------------------------------------------------------------------------
    def folded_kernel(self, name):
        v = self.params[f"{name}/v"].astype(np.float64)
        norm = np.sqrt((v ** 2).sum(axis=tuple(range(v.ndim - 1)), keepdims=True))
        return (self.params[f"{name}/g"].astype(np.float64) * v / norm).astype(np.float32)

    def pack(self):
        """Flat device buffer in the layout of include/inrhip.h (every segment padded to 4 floats)."""
        if self._packed is not None:
            return self._packed
        segs = []

        def put(a):
            a = np.ascontiguousarray(a, np.float32).reshape(-1)
            pad = (-a.size) % 4
            segs.append(np.concatenate([a, np.zeros(pad, np.float32)]) if pad else a)

        for name, ks, cin, cout in self.specs:
            w = self.folded_kernel(name).reshape(-1, cin, cout)        # TF order is already [tap][cin][cout]
            b = self.params[f"{name}/b"]
            if name == "up":                                           # zero-pad the output channels to the tile width
                w = np.concatenate([w, np.zeros(w.shape[:2] + (32 - cout,), np.float32)], axis=2)
                b = np.concatenate([b, np.zeros(32 - cout, np.float32)])
            if name == "stem":
                w = w.reshape(27, cout)
            put(w)
            put(b)
        flat = np.concatenate(segs)
        want = lib().inr_rams_param_count(C.byref(self.desc))
        if want < 0:
            check(int(want), "inr_rams_param_count")
        if flat.size != want:
            raise RuntimeError(f"packed RAMS parameters have {flat.size} floats, the library expects {want}")
        ops.require_gpu()
        self._packed = torch.from_numpy(flat).cuda()
        return self._packed

    def invalidate(self):
        self._packed = None

    # ---- weights on disk (the shipped TensorFlow checkpoints are incomplete -- the *.data-00001-of-00002 shard is stripped -- and
    #      there is no TensorFlow here to read one: weights travel as an .npz of the reference's variables, `<layer>/v|g|b`) -------------
    def save_weights(self, path):
        np.savez(path, **{k: np.asarray(v, np.float32) for k, v in self.params.items()})
        return path

    def load_weights(self, path):
        """Takes every ``<layer>/v``, ``<layer>/g``, ``<layer>/b`` of this architecture from an ``.npz`` (shapes checked)."""
        with np.load(path, allow_pickle=False) as z:
            new = {}
            for name, ks, cin, cout in self.specs:
                for suffix, shape in (("v", tuple(ks) + (cin, cout)), ("g", (cout,)), ("b", (cout,))):
                    key = f"{name}/{suffix}"
                    if key not in z.files:
                        raise KeyError(f"{path}: no array '{key}' (expected {len(self.specs) * 3} arrays named <layer>/v|g|b)")
                    a = np.asarray(z[key], np.float32)
                    if a.shape != shape:
                        raise ValueError(f"{path}: '{key}' has shape {a.shape}, this network needs {shape}")
                    new[key] = a
        self.params = new
        self.invalidate()
        return self

    # ---- forward ---------------------------------------------------------------------------------------------
    def forward(self, x, clip_round=False):
        """x: ``[B, H, W, channels]`` (numpy or tensor, any real dtype; cast to fp32 like prediction.py:78).
        Returns a device tensor ``[B, scale*H, scale*W, 1]``."""
        dev = ops.require_gpu()
        xt = torch.as_tensor(np.asarray(x, np.float32) if not torch.is_tensor(x) else x).to(dev, torch.float32).contiguous()
        if xt.dim() != 4 or xt.shape[-1] != self.cfg["channels"]:
            raise ValueError(f"input must be [B, H, W, {self.cfg['channels']}], got {tuple(xt.shape)}")
        B, H, W, _ = xt.shape
        s = self.cfg["scale"]
        out = torch.empty((B, H * s, W * s, 1), dtype=torch.float32, device=dev)
        nbytes = lib().inr_rams_workspace_bytes(C.byref(self.desc), B, H, W)
        if nbytes == 0:
            check(-1, "inr_rams_workspace_bytes")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        check(lib().inr_rams_forward(C.byref(self.desc), self.pack().data_ptr(), xt.data_ptr(), out.data_ptr(), B, H, W,
                                     1 if clip_round else 0, ws.data_ptr(), ws.numel(), ops._stream()),
              "inr_rams_forward")
        return out

    __call__ = forward


def predict_tensor(model: RAMS, x):
    """utils/prediction.py:76-83: cast to fp32 -> model -> clip to [0, 2**16] -> round (half to even)."""
    return model.forward(x, clip_round=True)


def predict_case(model: RAMS, stack, sample_size=25, rng=None):
    """multi-image-super-resolution/master.py:38-52: ``stack`` is one slice ``[H, W, T]`` of the DWI series; it is cast
    to uint16, multiplied by 256, ``sample_size`` random 9-acquisition subsets are super-resolved and averaged.
    The subsets run as one batch (the per-subset results are independent)."""
    import random
    rng = rng or random
    lor = np.asarray(stack)[None].astype("uint16") * 256
    T = lor.shape[3]
    ch = model.cfg["channels"]
    subsets = [rng.sample(list(range(T)), ch) for _ in range(sample_size)]
    batch = np.concatenate([lor[:, :, :, inx] for inx in subsets], axis=0).astype(np.float32)
    sr = predict_tensor(model, batch)[..., 0]
    return sr.double().mean(dim=0), subsets


# ---- RAMS+ (utils/prediction.py:10-74, 86-97): test-time ensembling.  Tensor plumbing around the forward pass -----------------------
def _as_dev(x):
    dev = ops.require_gpu()
    return torch.as_tensor(np.asarray(x, np.float32) if not torch.is_tensor(x) else x).to(dev, torch.float32)


def flip(X, rn=None, rng=None):
    """prediction.py:54-58: ``rn <= 0.5`` keeps X, otherwise ``tf.image.flip_left_right`` (the W axis: -2); returns
    (tensor, rint(rn)).  ``rn=None`` draws U[0, 1) from ``rng`` (a ``numpy.random.Generator``; the reference asks TensorFlow's)."""
    if rn is None:
        rn = float((rng or np.random.default_rng()).random())
    X = _as_dev(X)
    return (X if rn <= 0.5 else torch.flip(X, dims=(-2,))), float(np.rint(rn))


def rotate(X, rn=None, rng=None):
    """prediction.py:61-65: ``tf.image.rot90(X, rn)`` -- counter-clockwise quarter turns of the (H, W) = (-3, -2) axes."""
    if rn is None:
        rn = int((rng or np.random.default_rng()).integers(0, 4))
    X = _as_dev(X)
    return torch.rot90(X, int(rn) % 4, dims=(-3, -2)), rn


def shuffle_last_axis(X, rng=None):
    """prediction.py:68-73: transpose, shuffle the first axis, transpose back = a random permutation of the LAST axis (the
    acquisitions).  The permutation comes from ``rng`` (numpy) -- the reference's comes from TensorFlow's global generator."""
    X = _as_dev(X)
    perm = (rng or np.random.default_rng()).permutation(X.shape[-1])
    return X[..., torch.as_tensor(perm, device=X.device)]


def geometric_ensemble(X, shuffle=False, rng=None):
    """prediction.py:32-42: the eight flip x rotation images of one stack [H, W, T] -> ([8, ...], r [8, 2] = (flip, quarter turns))."""
    r = np.array(np.meshgrid([0, 1], [0, 1, 2, 3])).T.reshape(-1, 2)
    out = []
    for i in range(8):
        a = rotate(flip(X, r[i, 0])[0], r[i, 1])[0]
        out.append(shuffle_last_axis(a, rng) if shuffle else a)
    return torch.stack(out), r


def random_ensemble(X, n=10, shuffle=True, rng=None):
    """prediction.py:18-29: n random (flip, rotation) images."""
    rng = rng or np.random.default_rng()
    r = np.zeros((n, 2))
    out = []
    for i in range(n):
        a, r[i, 0] = flip(X, rng=rng)
        a, r[i, 1] = rotate(a, rng=rng)
        out.append(shuffle_last_axis(a, rng) if shuffle else a)
    return torch.stack(out), r


def ensemble(X, geometric=True, shuffle=False, n=10, rng=None):
    """prediction.py:10-15 (whose random branch forgets its ``return``: here it returns)."""
    return geometric_ensemble(X, shuffle, rng) if geometric else random_ensemble(X, n, shuffle, rng)


def unensemble(X, r):
    """prediction.py:45-51: undo rotation (k2 = 4 - k1) and flip of every member, mean over the members (kept as axis 0)."""
    X = _as_dev(X)
    members = [flip(rotate(X[i], 4 - int(r[i, 1]))[0], r[i, 0])[0] for i in range(len(X))]
    return torch.stack(members).double().mean(dim=0, keepdim=True)


def predict_tensor_permute(model: RAMS, x, n_ens=10, rng=None):
    """prediction.py:86-97: ``n_ens`` forward passes of ONE stack [H, W, T] with its acquisitions permuted, clipped and rounded,
    averaged -> [1, sH, sW, 1].  The permuted stacks run as one batch."""
    rng = rng or np.random.default_rng()
    batch = torch.stack([shuffle_last_axis(x, rng) for _ in range(n_ens)])
    return predict_tensor(model, batch).double().mean(dim=0, keepdim=True)


def _shift_loss(y_true, y_pred, y_mask, size, mode):
    dev = ops.require_gpu()

    def prep(t):
        t = torch.as_tensor(np.asarray(t, np.float32) if not torch.is_tensor(t) else t).to(dev, torch.float32)
        if t.dim() == 4 and t.shape[-1] == 1:
            t = t[..., 0]
        if t.dim() != 3 or t.shape[1] != size or t.shape[2] != size:
            raise ValueError(f"expected [B, {size}, {size}(, 1)], got {tuple(t.shape)}")
        return t.contiguous()

    yt, yp, mk = prep(y_true), prep(y_pred), prep(y_mask)
    if not (yt.shape == yp.shape == mk.shape):
        raise ValueError("y_true / y_pred / y_mask shape mismatch")
    B = yt.shape[0]
    out = torch.empty(B, dtype=torch.float64, device=dev)
    ws = torch.empty(lib().inr_rams_shift_loss_workspace_bytes(B, 3), dtype=torch.uint8, device=dev)
    check(lib().inr_rams_shift_loss(out.data_ptr(), yt.data_ptr(), yp.data_ptr(), mk.data_ptr(), B, int(size), 3, mode,
                                    ws.data_ptr(), ws.numel(), ops._stream()), "inr_rams_shift_loss")
    return out


def l1_loss(y_true, y_pred, y_mask, HR_SIZE=384):
    """utils/loss.py:26-75: per image, the minimum over the 7x7 label shifts of the brightness-corrected masked L1."""
    return _shift_loss(y_true, y_pred, y_mask, HR_SIZE, 0)


def psnr(y_true, y_pred, y_mask, size_image=384):
    """utils/loss.py:77-127: mean over the batch of the maximum cPSNR over the 7x7 shifts (data range 65535)."""
    return _shift_loss(y_true, y_pred, y_mask, size_image, 1).mean()


def l1_loss_and_grad(y_true, y_pred, y_mask, HR_SIZE=384, upstream=None):
    """The loss half of ``Trainer.train_step`` (utils/training.py:193-209): per-image cL1 (utils/loss.py:26-75) and the
    gradient of ``sum_b upstream[b] * loss[b]`` with respect to ``y_pred`` ([B, S, S] fp32), through the best shift."""
    dev = ops.require_gpu()

    def prep(t):
        t = torch.as_tensor(np.asarray(t, np.float32) if not torch.is_tensor(t) else t).to(dev, torch.float32)
        if t.dim() == 4 and t.shape[-1] == 1:
            t = t[..., 0]
        if t.dim() != 3 or t.shape[1] != HR_SIZE or t.shape[2] != HR_SIZE:
            raise ValueError(f"expected [B, {HR_SIZE}, {HR_SIZE}(, 1)], got {tuple(t.shape)}")
        return t.contiguous()

    yt, yp, mk = prep(y_true), prep(y_pred), prep(y_mask)
    B = yt.shape[0]
    up = None if upstream is None else torch.as_tensor(upstream, dtype=torch.float32, device=dev).contiguous()
    loss = torch.empty(B, dtype=torch.float64, device=dev)
    grad = torch.empty_like(yp)
    ws = torch.empty((lib().inr_rams_shift_loss_grad_workspace_bytes(B, 3) + 7) // 8, dtype=torch.float64, device=dev)
    check(lib().inr_rams_shift_loss_grad(loss.data_ptr(), grad.data_ptr(), yt.data_ptr(), yp.data_ptr(), mk.data_ptr(),
                                         0 if up is None else up.data_ptr(), B, int(HR_SIZE), 3, ws.data_ptr(), ws.numel() * 8,
                                         ops._stream()), "inr_rams_shift_loss_grad")
    return loss, grad


# ---- building blocks of the training step: the 3x3x3 convolution 32 -> 32 with its two gradients -----------------------
def _conv_tensors(x, w, name):
    dev = ops.require_gpu()
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5 and x.shape[-1] == 32 and x.is_contiguous()):
        raise ValueError(f"{name}: activations must be contiguous CUDA float32 [B, D1, D2, D3, 32]")
    if w is not None and not (w.is_cuda and w.dtype == torch.float32 and tuple(w.shape) == (27, 32, 32) and w.is_contiguous()):
        raise ValueError(f"{name}: w must be a contiguous CUDA float32 [27, 32, 32] (tap, cin, cout)")
    return dev


def conv3d(x, w, bias, pad=1, relu=False):
    """y = conv3d(x, w) + bias (optionally ReLU); x [B, D1, D2, D3, 32], w [27, 32, 32] folded kernel, pad 1 = 'same'."""
    dev = _conv_tensors(x, w, "conv3d")
    B, D1, D2, D3, _ = x.shape
    y = torch.empty((B, D1 + 2 * pad - 2, D2 + 2 * pad - 2, D3 + 2 * pad - 2, 32), dtype=torch.float32, device=dev)
    check(lib().inr_rams_conv3d_forward(y.data_ptr(), x.data_ptr(), w.data_ptr(), bias.contiguous().data_ptr(), B, D1, D2, D3,
                                        int(pad), 1 if relu else 0, ops._stream()), "inr_rams_conv3d_forward")
    return y


def conv3d_dgrad(dy, w):
    """d loss / d x of a 'same' convolution: the forward kernel on the flipped, channel-transposed kernel."""
    dev = _conv_tensors(dy, w, "conv3d_dgrad")
    B, D1, D2, D3, _ = dy.shape
    dx = torch.empty_like(dy)
    ws = torch.empty(lib().inr_rams_conv3d_dgrad_workspace_bytes() // 4, dtype=torch.float32, device=dev)
    check(lib().inr_rams_conv3d_dgrad(dx.data_ptr(), dy.data_ptr(), w.data_ptr(), B, D1, D2, D3, ws.data_ptr(), ws.numel() * 4,
                                      ops._stream()), "inr_rams_conv3d_dgrad")
    return dx


def conv3d_wgrad(x, dy, pad=1):
    """(d loss / d w [27, 32, 32], d loss / d bias [32]) of y = conv3d(x, w) + bias given dy."""
    dev = _conv_tensors(x, None, "conv3d_wgrad")
    _conv_tensors(dy, None, "conv3d_wgrad")
    B, D1, D2, D3, _ = x.shape
    if tuple(dy.shape) != (B, D1 + 2 * pad - 2, D2 + 2 * pad - 2, D3 + 2 * pad - 2, 32):
        raise ValueError("conv3d_wgrad: dy does not have the output shape of this convolution")
    gw = torch.empty((27, 32, 32), dtype=torch.float32, device=dev)
    gb = torch.empty(32, dtype=torch.float32, device=dev)
    ws = torch.empty(lib().inr_rams_conv3d_wgrad_workspace_bytes(B, D1, D2, D3, int(pad)) // 4 + 4, dtype=torch.float32, device=dev)
    check(lib().inr_rams_conv3d_wgrad(gw.data_ptr(), gb.data_ptr(), x.data_ptr(), dy.data_ptr(), B, D1, D2, D3, int(pad),
                                      ws.data_ptr(), ws.numel() * 4, ops._stream()), "inr_rams_conv3d_wgrad")
    return gw, gb


# ---- the training step (utils/training.py:193-209) --------------------------------------------------------------------------
class RamsTrainer:
    """``Trainer.train_step`` of the reference (utils/training.py:193-209) on the device: forward with every intermediate
    kept -> per-image cL1 (utils/loss.py:26-75) -> gradients of ``sum_b loss_b`` (what ``tape.gradient`` of the loss vector
    is) with respect to every ``v``, ``g`` and bias -> Adam in Keras' form.  One C call per step (``inr_rams_train_step``).

    The model's parameters live in one flat device buffer in the reference's variables (per layer ``v``, ``g``, ``b``:
    the "raw" layout of include/inrhip.h); ``sync_model()`` writes them back into ``model.params`` (and drops the model's
    packed inference copy)."""

    def __init__(self, model: RAMS, learning_rate=5e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        dev = ops.require_gpu()
        self.model = model
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), float(beta_1), float(beta_2), float(epsilon)
        total = lib().inr_rams_train_param_count(C.byref(model.desc))
        if total < 0:
            check(int(total), "inr_rams_train_param_count")
        offs = (C.c_int64 * (3 * 128))()
        n = lib().inr_rams_train_param_offsets(C.byref(model.desc), offs, 128)
        if n != len(model.specs):
            raise RuntimeError(f"library reports {n} layers, the model has {len(model.specs)}")
        self.offsets = [(int(offs[3 * i]), int(offs[3 * i + 1]), int(offs[3 * i + 2])) for i in range(n)]
        flat = np.zeros(int(total), np.float32)
        for (name, ks, cin, cout), (ov, og, ob) in zip(model.specs, self.offsets):
            v = np.asarray(model.params[f"{name}/v"], np.float32).reshape(-1)
            flat[ov:ov + v.size] = v
            flat[og:og + cout] = model.params[f"{name}/g"]
            flat[ob:ob + cout] = model.params[f"{name}/b"]
        self.flat = torch.from_numpy(flat).to(dev)
        self.grads = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.step_count = 0
        self._ws = None

    def _prep(self, lr_batch, hr, mask):
        dev = self.flat.device

        def as_dev(t, squeeze):
            t = torch.as_tensor(np.asarray(t, np.float32) if not torch.is_tensor(t) else t).to(dev, torch.float32)
            if squeeze and t.dim() == 4 and t.shape[-1] == 1:
                t = t[..., 0]
            return t.contiguous()

        x = as_dev(lr_batch, False)                     # tf.cast(lr, tf.float32), training.py:195
        yt, mk = as_dev(hr, True), as_dev(mask, True)
        B, H, W, ch = x.shape
        s = self.model.cfg["scale"]
        if ch != self.model.cfg["channels"] or tuple(yt.shape) != (B, H * s, W * s) or tuple(mk.shape) != tuple(yt.shape):
            raise ValueError(f"shapes: lr {tuple(x.shape)}, hr {tuple(yt.shape)}, mask {tuple(mk.shape)}")
        need = lib().inr_rams_train_workspace_bytes(C.byref(self.model.desc), B, H, W)
        if need == 0:
            check(-1, "inr_rams_train_workspace_bytes")
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        return x, yt, mk, B, H, W

    def loss_and_grads(self, lr_batch, hr, mask, want_prediction=False):
        """Per-image loss [B] (float64) and ``d sum(loss) / d parameters`` (left in ``self.grads``); no update."""
        x, yt, mk, B, H, W = self._prep(lr_batch, hr, mask)
        s = self.model.cfg["scale"]
        loss = torch.empty(B, dtype=torch.float64, device=x.device)
        pred = torch.empty((B, H * s, W * s), dtype=torch.float32, device=x.device) if want_prediction else None
        check(lib().inr_rams_train_grads(C.byref(self.model.desc), self.flat.data_ptr(), self.grads.data_ptr(), x.data_ptr(),
                                         yt.data_ptr(), mk.data_ptr(), loss.data_ptr(), 0 if pred is None else pred.data_ptr(),
                                         B, H, W, self._ws.data_ptr(), self._ws.numel(), ops._stream()), "inr_rams_train_grads")
        return (loss, pred) if want_prediction else loss

    def train_step(self, lr_batch, hr, mask):
        """One optimizer step; returns the per-image loss vector the reference feeds to ``train_loss`` (training.py:207)."""
        x, yt, mk, B, H, W = self._prep(lr_batch, hr, mask)
        loss = torch.empty(B, dtype=torch.float64, device=x.device)
        self.step_count += 1
        check(lib().inr_rams_train_step(C.byref(self.model.desc), self.flat.data_ptr(), self.grads.data_ptr(), self.m.data_ptr(),
                                        self.v.data_ptr(), x.data_ptr(), yt.data_ptr(), mk.data_ptr(), loss.data_ptr(), B, H, W,
                                        self.step_count, self.lr, self.b1, self.b2, self.eps, self._ws.data_ptr(),
                                        self._ws.numel(), ops._stream()), "inr_rams_train_step")
        self.model.invalidate()
        return loss

    def named_gradients(self):
        """{``<layer>/v|g|b``: ndarray} views of the last gradient, in the reference's variable shapes."""
        g = self.grads.cpu().numpy()
        return self._unflatten(g)

    def _unflatten(self, flat):
        out = {}
        for (name, ks, cin, cout), (ov, og, ob) in zip(self.model.specs, self.offsets):
            n = int(np.prod(ks)) * cin * cout
            out[f"{name}/v"] = flat[ov:ov + n].reshape(tuple(ks) + (cin, cout)).copy()
            out[f"{name}/g"] = flat[og:og + cout].copy()
            out[f"{name}/b"] = flat[ob:ob + cout].copy()
        return out

    def sync_model(self):
        self.model.params = self._unflatten(self.flat.cpu().numpy())
        self.model.invalidate()
        return self.model

    # ---- the outer loop (utils/training.py:108-191, 211-220) ---------------------------------------------------------------------
    def test_step(self, lr_batch, hr, mask):
        """training.py:211-220: forward (no update), per-image cL1 and the batch's cPSNR."""
        size = int(np.asarray(hr).shape[1]) if not torch.is_tensor(hr) else int(hr.shape[1])
        sr = self.sync_model().forward(lr_batch)
        return l1_loss(hr, sr, mask, HR_SIZE=size), psnr(hr, sr, mask, size_image=size)

    def save_checkpoint(self, directory, psnr_value, max_to_keep=3):
        """``tf.train.CheckpointManager(max_to_keep=3).save()`` (training.py:88-91, 187): step, best PSNR, variables, Adam state."""
        import glob
        import os
        os.makedirs(directory, exist_ok=True)
        path = os.path.join(directory, f"ckpt-{self.step_count:08d}.npz")
        np.savez(path, step=self.step_count, psnr=float(psnr_value), flat=self.flat.cpu().numpy(), m=self.m.cpu().numpy(),
                 v=self.v.cpu().numpy())
        for old in sorted(glob.glob(os.path.join(directory, "ckpt-*.npz")))[:-max_to_keep]:
            os.remove(old)
        return path

    def restore(self, directory):
        """training.py:93-97: continue from the latest checkpoint of `directory` if there is one; returns its best PSNR or None."""
        import glob
        import os
        found = sorted(glob.glob(os.path.join(directory, "ckpt-*.npz")))
        if not found:
            return None
        z = np.load(found[-1])
        dev = self.flat.device
        self.flat.copy_(torch.from_numpy(z["flat"]).to(dev))
        self.m.copy_(torch.from_numpy(z["m"]).to(dev))
        self.v.copy_(torch.from_numpy(z["v"]).to(dev))
        self.step_count = int(z["step"])
        self.model.invalidate()
        return float(z["psnr"])

    def fit(self, x, y, batch_size, epochs=100, evaluate_every=100, val_steps=100, validation_data=None, shuffle=True,
            initial_epoch=0, save_best_only=True, checkpoint_dir=None, seed=0, log=None):
        """``Trainer.fit`` (training.py:108-191): ``x`` [n, h, w, T] low-resolution stacks, ``y = (hr, mask)``; per epoch the data in
        a fresh random order (the reference: ``tf.data`` shuffle with a 512-element buffer, reshuffled each iteration -- a uniform
        permutation here, seeded), batches of ``batch_size`` (the last one may be short), one ``train_step`` each; every
        ``evaluate_every`` steps ``test_step`` over at most ``val_steps`` validation batches, and a checkpoint when the mean
        validation cPSNR improved (or always with ``save_best_only=False``).  Construction-time restore as the reference:
        ``checkpoint_dir`` holding checkpoints continues from the latest.  Returns the history (one record per evaluation).
        TensorBoard writers and the progress bar of the reference are replaced by ``log(record)``."""
        hr, mask = y
        n = len(x)
        rng = np.random.default_rng(seed)
        best = 1.0                                            # training.py:86: psnr = tf.Variable(1.0)
        if checkpoint_dir:
            got = self.restore(checkpoint_dir)
            best = best if got is None else got
        history, train_losses = [], []
        for epoch in range(epochs - initial_epoch):
            order = rng.permutation(n) if shuffle else np.arange(n)
            for b0 in range(0, n, batch_size):
                idx = order[b0:b0 + batch_size]
                train_losses.append(self.train_step(x[idx], hr[idx], mask[idx]).mean())
                if self.step_count % evaluate_every:
                    continue
                rec = {"epoch": epoch + 1 + initial_epoch, "step": self.step_count,
                       "loss": float(torch.stack(train_losses).mean())}
                train_losses = []
                if validation_data is not None:
                    vx, (vhr, vmask) = validation_data
                    vorder = rng.permutation(len(vx))
                    vl, vp = [], []
                    for k in range(0, min(len(vx), val_steps * batch_size), batch_size):
                        vi = vorder[k:k + batch_size]
                        l, p = self.test_step(vx[vi], vhr[vi], vmask[vi])
                        vl.append(l.mean())
                        vp.append(p)
                    rec["val_loss"], rec["val_psnr"] = float(torch.stack(vl).mean()), float(torch.stack(vp).mean())
                    improved = rec["val_psnr"] > best
                    if checkpoint_dir and (improved or not save_best_only):
                        best = rec["val_psnr"]                # (training.py:186: assigned whenever a checkpoint is written)
                        rec["checkpoint"] = self.save_checkpoint(checkpoint_dir, best)
                history.append(rec)
                if log:
                    log(rec)
        return history
