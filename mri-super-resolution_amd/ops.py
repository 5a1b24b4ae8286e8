"""torch-tensor wrappers over the C ABI (include/inrhip.h).

PyTorch is used here only for device memory (allocation of outputs/workspaces) and for the
current HIP stream.  Every wrapper validates device, dtype, contiguity and shapes on the host
BEFORE a kernel is enqueued, and raises instead of falling back: tensors must live on a HIP
device and libinrhip.so must be loadable.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import SirenDesc, check, lib, shape_array


class InrDeviceError(RuntimeError):
    """Raised when an op is asked to run without a HIP device (there is no CPU path)."""


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise InrDeviceError("mri-super-resolution_amd needs a HIP GPU (MI355X / gfx950); "
                             "no CPU fallback exists for the INR kernels.")
    lib()
    return torch.device("cuda", torch.cuda.current_device())


def _chk(t: torch.Tensor, name: str, shape=None) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise InrDeviceError(f"{name} is on {t.device}: the INR kernels only run on a HIP device "
                             "(move it with .cuda(); there is no CPU fallback)")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def make_desc(in_features, hidden_features, hidden_layers, out_features, first_omega=30.0, hidden_omega=30.0):
    return SirenDesc(int(in_features), int(hidden_features), int(hidden_layers), int(out_features),
                     float(first_omega), float(hidden_omega))


def device_caps(device: int = 0) -> dict:
    caps = _lib.DeviceCaps()
    check(lib().inr_device_caps(int(device), C.byref(caps)), "inr_device_caps")
    return {f: (getattr(caps, f).decode() if f == "arch" else getattr(caps, f)) for f, _ in caps._fields_}


# ---- a-1 / a-3 ------------------------------------------------------------------------------------
def mgrid(shape, row_begin: int = 0, n_rows: int | None = None) -> torch.Tensor:
    dev = require_gpu()
    shape = tuple(int(s) for s in shape)
    total = 1
    for s in shape:
        total *= s
    n_rows = total - row_begin if n_rows is None else int(n_rows)
    if row_begin < 0 or n_rows < 0 or row_begin + n_rows > total:
        raise ValueError("row range outside the grid")
    out = torch.empty((n_rows, len(shape)), dtype=torch.float32, device=dev)
    if n_rows == 0:
        return out
    check(lib().inr_mgrid(out.data_ptr(), shape_array(shape), len(shape), row_begin, n_rows, _stream()), "inr_mgrid")
    return out


def fourier_map(x: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    _chk(x, "x")
    _chk(B, "B")
    if x.dim() != 2 or B.dim() != 2 or x.shape[1] != B.shape[1]:
        raise ValueError(f"x {tuple(x.shape)} and B {tuple(B.shape)} must be [n,d] and [m,d]")
    n, d = x.shape
    m = B.shape[0]
    out = torch.empty((n, 2 * m), dtype=torch.float32, device=x.device)
    if n == 0:
        return out
    check(lib().inr_fourier_map(out.data_ptr(), x.data_ptr(), B.data_ptr(), n, d, m, _stream()), "inr_fourier_map")
    return out


def grid_fourier_map(shape, B: torch.Tensor, row_begin: int = 0, n_rows: int | None = None) -> torch.Tensor:
    _chk(B, "B")
    shape = tuple(int(s) for s in shape)
    if B.dim() != 2 or B.shape[1] != len(shape):
        raise ValueError(f"B {tuple(B.shape)} must be [m,{len(shape)}]")
    total = 1
    for s in shape:
        total *= s
    n_rows = total - row_begin if n_rows is None else int(n_rows)
    if row_begin < 0 or n_rows < 0 or row_begin + n_rows > total:
        raise ValueError("row range outside the grid")
    m = B.shape[0]
    out = torch.empty((n_rows, 2 * m), dtype=torch.float32, device=B.device)
    if n_rows == 0:
        return out
    check(lib().inr_grid_fourier_map(out.data_ptr(), shape_array(shape), len(shape), row_begin, n_rows,
                                     B.data_ptr(), m, _stream()), "inr_grid_fourier_map")
    return out


# ---- per-layer pieces (autograd path) -------------------------------------------------------------------
def sine_layer_forward(x, W, b, omega: float, stash: bool):
    _chk(x, "x")
    _chk(W, "weight")
    if x.dim() != 2 or W.dim() != 2 or x.shape[1] != W.shape[1]:
        raise ValueError(f"x {tuple(x.shape)} / weight {tuple(W.shape)} mismatch")
    n, fin = x.shape
    fout = W.shape[0]
    if b is not None:
        _chk(b, "bias", (fout,))
    act = torch.empty((n, fout), dtype=torch.float32, device=x.device)
    dact = torch.empty_like(act) if stash else None
    check(lib().inr_sine_layer_forward(act.data_ptr(), _ptr(dact), x.data_ptr(), W.data_ptr(), _ptr(b), n, fin, fout,
                                       float(omega), _stream()), "inr_sine_layer_forward")
    return act, dact


def tanh_layer_forward(x, W, b, scale: float, stash: bool):
    """act = scale*tanh(x W^T + b) and (if ``stash``) its derivative factor scale*(1-tanh^2)."""
    _chk(x, "x")
    _chk(W, "weight")
    if x.dim() != 2 or W.dim() != 2 or x.shape[1] != W.shape[1]:
        raise ValueError(f"x {tuple(x.shape)} / weight {tuple(W.shape)} mismatch")
    n, fin = x.shape
    fout = W.shape[0]
    if b is not None:
        _chk(b, "bias", (fout,))
    act = torch.empty((n, fout), dtype=torch.float32, device=x.device)
    dact = torch.empty_like(act) if stash else None
    check(lib().inr_tanh_layer_forward(act.data_ptr(), _ptr(dact), x.data_ptr(), W.data_ptr(), _ptr(b), n, fin, fout,
                                       float(scale), _stream()), "inr_tanh_layer_forward")
    return act, dact


def linear_tanh_head_forward(a, W, b, scale: float, stash: bool):
    """y = scale*tanh(a W^T + b) for a few output columns, and (if ``stash``) scale*(1-tanh^2)."""
    _chk(a, "a")
    _chk(W, "weight")
    if a.dim() != 2 or W.dim() != 2 or a.shape[1] != W.shape[1]:
        raise ValueError(f"a {tuple(a.shape)} / weight {tuple(W.shape)} mismatch")
    n, hidden = a.shape
    out_f = W.shape[0]
    if b is not None:
        _chk(b, "bias", (out_f,))
    y = torch.empty((n, out_f), dtype=torch.float32, device=a.device)
    dy = torch.empty_like(y) if stash else None
    check(lib().inr_linear_tanh_head_forward(y.data_ptr(), _ptr(dy), a.data_ptr(), W.data_ptr(), _ptr(b), n, hidden,
                                             out_f, float(scale), _stream()), "inr_linear_tanh_head_forward")
    return y, dy


def mul(a, b):
    _chk(a, "a")
    _chk(b, "b", a.shape)
    out = torch.empty_like(a)
    check(lib().inr_mul(out.data_ptr(), a.data_ptr(), b.data_ptr(), a.numel(), _stream()), "inr_mul")
    return out


def acquisition_products(raw_b0, raw_b1, raw_b2, raw_b3):
    """All combinations of one acquisition per b-value at every voxel (SRDWI.py:143-152 over superresDWI.py:57-76):
    raw_b0 [...], raw_bk [..., nk] device fp32 -> [..., 4, n1*n2*n3] in itertools.product order."""
    for name, t in (("raw_b0", raw_b0), ("raw_b1", raw_b1), ("raw_b2", raw_b2), ("raw_b3", raw_b3)):
        _chk(t, name)
    lead = tuple(raw_b0.shape)
    if any(tuple(t.shape[:-1]) != lead for t in (raw_b1, raw_b2, raw_b3)):
        raise ValueError("acquisition_products: leading (voxel) shapes differ")
    n1, n2, n3 = raw_b1.shape[-1], raw_b2.shape[-1], raw_b3.shape[-1]
    out = torch.empty(lead + (4, n1 * n2 * n3), dtype=torch.float32, device=raw_b0.device)
    check(lib().inr_acquisition_products(out.data_ptr(), raw_b0.data_ptr(), raw_b1.data_ptr(), raw_b2.data_ptr(),
                                         raw_b3.data_ptr(), raw_b0.numel(), n1, n2, n3, _stream()), "inr_acquisition_products")
    return out


def linear_head_forward(a, W, b, clamp_min=None):
    _chk(a, "a")
    _chk(W, "weight")
    if a.dim() != 2 or W.dim() != 2 or a.shape[1] != W.shape[1]:
        raise ValueError(f"a {tuple(a.shape)} / weight {tuple(W.shape)} mismatch")
    n, hidden = a.shape
    out_f = W.shape[0]
    if b is not None:
        _chk(b, "bias", (out_f,))
    y = torch.empty((n, out_f), dtype=torch.float32, device=a.device)
    check(lib().inr_linear_head_forward(y.data_ptr(), a.data_ptr(), W.data_ptr(), _ptr(b), n, hidden, out_f,
                                        0 if clamp_min is None else 1, float(clamp_min or 0.0), _stream()),
          "inr_linear_head_forward")
    return y


def mse_loss_grad(y, t, w=None):
    """Returns (loss[1] device tensor, gy like y)."""
    _chk(y, "y")
    _chk(t, "target", y.shape)
    if w is not None:
        _chk(w, "weight", y.shape)
    count = y.numel()
    gy = torch.empty_like(y)
    loss = torch.empty((1,), dtype=torch.float32, device=y.device)
    nbytes = lib().inr_mse_workspace_bytes(count)
    ws = _ws(nbytes, y.device)
    check(lib().inr_mse_loss_grad(gy.data_ptr(), loss.data_ptr(), y.data_ptr(), t.data_ptr(), _ptr(w), count,
                                  ws.data_ptr(), ws.numel(), _stream()), "inr_mse_loss_grad")
    return loss, gy


def linear_head_backward(gy, a_last, dact_last, W, need_dz=True, need_param=True, need_bias_last=False):
    """dz_last = (gy W) * dact_last (plain product when dact_last is None); gW, gb of the head; with
    ``need_bias_last`` also colsum(dz_last), the bias gradient of the last sine layer."""
    _chk(gy, "gy")
    _chk(a_last, "a_last")
    _chk(W, "weight")
    n, hidden = a_last.shape
    out_f = W.shape[0]
    if tuple(gy.shape) != (n, out_f) or tuple(W.shape) != (out_f, hidden):
        raise ValueError("head backward shape mismatch")
    if dact_last is not None:
        _chk(dact_last, "dact_last", (n, hidden))
    dz = torch.empty((n, hidden), dtype=torch.float32, device=gy.device) if need_dz else None
    gW = torch.empty_like(W) if need_param else None
    gb = torch.empty((out_f,), dtype=torch.float32, device=gy.device) if need_param else None
    gb_last = torch.empty((hidden,), dtype=torch.float32, device=gy.device) if (need_bias_last and need_dz) else None
    nbytes = lib().inr_head_backward_workspace_bytes(n, hidden, out_f)
    ws = _ws(nbytes, gy.device)
    check(lib().inr_linear_head_backward(_ptr(dz), _ptr(gW), _ptr(gb), _ptr(gb_last), gy.data_ptr(), a_last.data_ptr(),
                                         _ptr(dact_last), W.data_ptr(), n, hidden, out_f, ws.data_ptr(), ws.numel(),
                                         _stream()), "inr_linear_head_backward")
    return dz, gW, gb, gb_last


def sine_layer_backward_input(dz, W, dact_prev, out=None, need_bias=False):
    """(dz @ W) * dact_prev  ->  [n, in]; `out` may alias dact_prev (in place).  With ``need_bias`` also returns
    colsum of the result (bias gradient of the layer below) from the fused epilogue."""
    _chk(dz, "dz")
    _chk(W, "weight")
    n, fout = dz.shape
    if W.shape[0] != fout:
        raise ValueError("dz / weight mismatch")
    fin = W.shape[1]
    if dact_prev is not None:
        _chk(dact_prev, "dact_prev", (n, fin))
    if out is None:
        out = torch.empty((n, fin), dtype=torch.float32, device=dz.device)
    else:
        _chk(out, "out", (n, fin))
    gb = ws = None
    ws_ptr, ws_bytes = 0, 0
    if need_bias:
        gb = torch.empty((fin,), dtype=torch.float32, device=dz.device)
        ws = _ws(lib().inr_sine_layer_backward_input_workspace_bytes(n, fin), dz.device)
        ws_ptr, ws_bytes = ws.data_ptr(), ws.numel()
    check(lib().inr_sine_layer_backward_input(out.data_ptr(), _ptr(gb), dz.data_ptr(), W.data_ptr(), _ptr(dact_prev), n,
                                              fin, fout, ws_ptr, ws_bytes, _stream()),
          "inr_sine_layer_backward_input")
    return (out, gb) if need_bias else out


def linear_param_grad(dz, x, need_bias=True):
    _chk(dz, "dz")
    _chk(x, "x")
    n, fout = dz.shape
    if x.shape[0] != n:
        raise ValueError("dz / x row mismatch")
    fin = x.shape[1]
    gW = torch.empty((fout, fin), dtype=torch.float32, device=dz.device)
    gb = torch.empty((fout,), dtype=torch.float32, device=dz.device) if need_bias else None
    nbytes = lib().inr_linear_param_grad_workspace_bytes(n, fin, fout)
    ws = _ws(nbytes, dz.device)
    check(lib().inr_linear_param_grad(gW.data_ptr(), _ptr(gb), dz.data_ptr(), x.data_ptr(), n, fin, fout,
                                      ws.data_ptr(), ws.numel(), _stream()), "inr_linear_param_grad")
    return gW, gb


def adam_step(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
    for name, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _chk(t, name)
        if t.numel() != p.numel():
            raise ValueError("adam tensors must have equal numel")
    check(lib().inr_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), int(step),
                              float(lr), float(beta1), float(beta2), float(eps), _stream()), "inr_adam_step")


# ---- fused SIREN -----------------------------------------------------------------------------------------
def siren_param_layout(desc: SirenDesc):
    """(total_floats, [(w_off, b_off)] per layer in network order, head last)."""
    total = lib().inr_siren_param_count(C.byref(desc))
    if total < 0:
        check(int(total), "inr_siren_param_count")
    n_layers = desc.hidden_layers + 2
    offs = (C.c_int64 * (2 * n_layers))()
    check(lib().inr_siren_param_offsets(C.byref(desc), offs), "inr_siren_param_offsets")
    return int(total), [(int(offs[2 * l]), int(offs[2 * l + 1])) for l in range(n_layers)]


def siren_forward(desc: SirenDesc, params, x, clamp_min=None):
    _chk(params, "params")
    _chk(x, "x")
    if x.dim() != 2 or x.shape[1] != desc.in_features:
        raise ValueError(f"x {tuple(x.shape)} must be [n,{desc.in_features}]")
    total, _ = siren_param_layout(desc)
    if params.numel() != total:
        raise ValueError(f"flat params has {params.numel()} floats, layout needs {total}")
    n = x.shape[0]
    y = torch.empty((n, desc.out_features), dtype=torch.float32, device=x.device)
    nbytes = lib().inr_siren_forward_workspace_bytes(C.byref(desc), n)
    ws = _ws(nbytes, x.device)
    check(lib().inr_siren_forward(C.byref(desc), params.data_ptr(), x.data_ptr(), n, y.data_ptr(),
                                  0 if clamp_min is None else 1, float(clamp_min or 0.0), ws.data_ptr(), ws.numel(),
                                  _stream()), "inr_siren_forward")
    return y


def siren_reconstruct(desc: SirenDesc, params, shape, B=None, clamp_min=0.0, chunk_rows: int = 1 << 20):
    _chk(params, "params")
    shape = tuple(int(s) for s in shape)
    total_rows = 1
    for s in shape:
        total_rows *= s
    total, _ = siren_param_layout(desc)
    if params.numel() != total:
        raise ValueError(f"flat params has {params.numel()} floats, layout needs {total}")
    m = 0
    if B is not None:
        _chk(B, "B")
        if B.dim() != 2 or B.shape[1] != len(shape) or 2 * B.shape[0] != desc.in_features:
            raise ValueError(f"B {tuple(B.shape)} must be [{desc.in_features // 2},{len(shape)}]")
        m = B.shape[0]
    elif len(shape) != desc.in_features:
        raise ValueError("without B the grid dimension must equal in_features")
    chunk_rows = max(1, min(int(chunk_rows), total_rows))
    y = torch.empty((total_rows, desc.out_features), dtype=torch.float32, device=params.device)
    nbytes = lib().inr_siren_reconstruct_workspace_bytes(C.byref(desc), chunk_rows)
    ws = _ws(nbytes, params.device)
    check(lib().inr_siren_reconstruct(C.byref(desc), params.data_ptr(), shape_array(shape), len(shape), _ptr(B), m,
                                      y.data_ptr(), 0 if clamp_min is None else 1, float(clamp_min or 0.0),
                                      chunk_rows, ws.data_ptr(), ws.numel(), _stream()), "inr_siren_reconstruct")
    return y


def siren_fit_workspace_bytes(desc: SirenDesc, n: int) -> int:
    return int(lib().inr_siren_fit_workspace_bytes(C.byref(desc), int(n)))


def siren_fit(desc: SirenDesc, params, grads, m, v, x, target, weight, first_step: int, n_steps: int, lr: float,
              beta1=0.9, beta2=0.999, eps=1e-8, losses=None, workspace=None):
    total, _ = siren_param_layout(desc)
    for name, t in (("params", params), ("grads", grads), ("m", m), ("v", v)):
        _chk(t, name)
        if t.numel() != total:
            raise ValueError(f"{name} has {t.numel()} floats, layout needs {total}")
    _chk(x, "x")
    if x.dim() != 2 or x.shape[1] != desc.in_features:
        raise ValueError(f"x {tuple(x.shape)} must be [n,{desc.in_features}]")
    n = x.shape[0]
    _chk(target, "target")
    if target.numel() != n * desc.out_features:
        raise ValueError("target must have n*out_features elements")
    if weight is not None:
        _chk(weight, "weight")
        if weight.numel() != n * desc.out_features:
            raise ValueError("weight must have n*out_features elements")
    if losses is not None:
        _chk(losses, "losses")
        if losses.numel() < n_steps:
            raise ValueError("losses buffer shorter than n_steps")
    need = siren_fit_workspace_bytes(desc, n)
    if workspace is None:
        workspace = _ws(need, x.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError("workspace too small")
    check(lib().inr_siren_fit(C.byref(desc), params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(),
                              x.data_ptr(), target.data_ptr(), _ptr(weight), n, int(first_step), int(n_steps),
                              float(lr), float(beta1), float(beta2), float(eps), _ptr(losses), workspace.data_ptr(),
                              workspace.numel() * workspace.element_size(), _stream()), "inr_siren_fit")
    return workspace


def siren_fit_cycle(desc: SirenDesc, params, grads, m, v, x, targets, weights, first_acq: int, first_step: int,
                    n_steps: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8, losses=None, workspace=None):
    """``inr_siren_fit_cycle``: ``targets`` (and ``weights``) are [n_acq, n * out_features]; step ``it`` fits acquisition
    ``(first_acq + it) % n_acq`` (master.py:137-148)."""
    total, _ = siren_param_layout(desc)
    for name, t in (("params", params), ("grads", grads), ("m", m), ("v", v)):
        _chk(t, name)
        if t.numel() != total:
            raise ValueError(f"{name} has {t.numel()} floats, layout needs {total}")
    _chk(x, "x")
    if x.dim() != 2 or x.shape[1] != desc.in_features:
        raise ValueError(f"x {tuple(x.shape)} must be [n,{desc.in_features}]")
    n = x.shape[0]
    _chk(targets, "targets")
    if targets.dim() != 2 or targets.shape[1] != n * desc.out_features:
        raise ValueError("targets must be [n_acq, n*out_features]")
    n_acq = targets.shape[0]
    if weights is not None:
        _chk(weights, "weights")
        if tuple(weights.shape) != tuple(targets.shape):
            raise ValueError("weights must have the shape of targets")
    if not 0 <= int(first_acq) < n_acq:
        raise ValueError("first_acq out of range")
    if losses is not None:
        _chk(losses, "losses")
        if losses.numel() < n_steps:
            raise ValueError("losses buffer shorter than n_steps")
    need = siren_fit_workspace_bytes(desc, n)
    if workspace is None:
        workspace = _ws(need, x.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError("workspace too small")
    check(lib().inr_siren_fit_cycle(C.byref(desc), params.data_ptr(), grads.data_ptr(), m.data_ptr(), v.data_ptr(),
                                    x.data_ptr(), targets.data_ptr(), _ptr(weights), int(n_acq), int(first_acq), n,
                                    int(first_step), int(n_steps), float(lr), float(beta1), float(beta2), float(eps),
                                    _ptr(losses), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                    _stream()), "inr_siren_fit_cycle")
    return workspace


REUSE_INPUT_IMAGE, REUSE_TARGET_STATS = 1, 2        # INR_REUSE_* of include/inrhip.h


def siren_loss_grad(desc: SirenDesc, params, grads, x, target, weight, count_total: int, loss, workspace=None, flags: int = 0):
    """Forward + loss + backward of one row shard (no optimizer); see ``inr_siren_loss_grad`` / ``_ex`` (``flags``: the
    caller vouches that x / the targets are those of the previous call on this workspace)."""
    total, _ = siren_param_layout(desc)
    for name, t in (("params", params), ("grads", grads)):
        _chk(t, name)
        if t.numel() != total:
            raise ValueError(f"{name} has {t.numel()} floats, layout needs {total}")
    _chk(x, "x")
    _chk(target, "target")
    _chk(loss, "loss")
    n = x.shape[0]
    if x.dim() != 2 or x.shape[1] != desc.in_features or target.numel() != n * desc.out_features:
        raise ValueError("x / target shape mismatch")
    if weight is not None:
        _chk(weight, "weight")
    need = siren_fit_workspace_bytes(desc, n)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        if flags:
            raise ValueError("REUSE_* flags need the workspace of the previous call")
        workspace = _ws(need, x.device)
    check(lib().inr_siren_loss_grad_ex(C.byref(desc), params.data_ptr(), grads.data_ptr(), x.data_ptr(), target.data_ptr(),
                                       _ptr(weight), n, int(count_total), loss.data_ptr(), workspace.data_ptr(),
                                       workspace.numel() * workspace.element_size(), int(flags), _stream()),
          "inr_siren_loss_grad_ex")
    return workspace


def siren_hp_eligible(desc: SirenDesc) -> bool:
    """May the whole-network entry points run this shape on the pre-split kernels (``inr_siren_forward_train`` needs it)?"""
    return bool(lib().inr_siren_hp_eligible(C.byref(desc)))


def siren_forward_train(desc: SirenDesc, params, x, workspace=None, flags: int = 0):
    """``inr_siren_forward_train``: y = network(x), every layer's stash left in ``workspace`` for ``siren_backward_train``.
    Returns (y [n, 1], workspace)."""
    total, _ = siren_param_layout(desc)
    _chk(params, "params")
    _chk(x, "x")
    if params.numel() != total or x.dim() != 2 or x.shape[1] != desc.in_features:
        raise ValueError("params / x shape mismatch")
    n = x.shape[0]
    need = siren_fit_workspace_bytes(desc, n)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        if flags:
            raise ValueError("REUSE_* flags need the workspace of the previous call")
        workspace = _ws(need, x.device)
    y = torch.empty(n, desc.out_features, dtype=torch.float32, device=x.device)
    check(lib().inr_siren_forward_train(C.byref(desc), params.data_ptr(), x.data_ptr(), y.data_ptr(), n, workspace.data_ptr(),
                                        workspace.numel() * workspace.element_size(), int(flags), _stream()),
          "inr_siren_forward_train")
    return y, workspace


def siren_backward_train(desc: SirenDesc, params, grads, gy, workspace):
    """``inr_siren_backward_train``: the flat gradient (network order) of sum(gy * y) from the pending forward's stash."""
    _chk(params, "params")
    _chk(grads, "grads")
    _chk(gy, "gy")
    n = gy.numel() // desc.out_features
    check(lib().inr_siren_backward_train(C.byref(desc), params.data_ptr(), grads.data_ptr(), gy.data_ptr(), n,
                                         workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()),
          "inr_siren_backward_train")
    return grads


# ---- diagnostics ------------------------------------------------------------------------------------------
LAUNCH_FAMILIES = ("hp_pkd", "hp_pkc", "hp_tile", "hp_rc", "h3", "f32_pipe16", "f32_pipe", "f32_generic", "small_multi",
                   "small_step", "hp_narrow", "hp_fused_fwd", "hp_row")   # INR_LF_* of include/inrhip.h, in order


def launch_counts() -> dict:
    """Launches each kernel family has received since the last ``launch_counts_reset()`` (process-global)."""
    out = {}
    for i, name in enumerate(LAUNCH_FAMILIES):
        n = C.c_int64(0)
        check(lib().inr_launch_count(i, C.byref(n)), "inr_launch_count")
        out[name] = int(n.value)
    return out


def launch_counts_reset():
    check(lib().inr_launch_counts_reset())


def debug_get(key: int) -> int:
    v = C.c_int(0)
    check(lib().inr_debug_get(int(key), C.byref(v)), "inr_debug_get")
    return int(v.value)


class debug_switch:
    """``with debug_switch(key, value): ...`` -- a diagnostic switch (process-global, include/inrhip.h) set for the block
    and put back to what it was, also when the block raises."""

    def __init__(self, key: int, value: int):
        self.key, self.value = int(key), int(value)

    def __enter__(self):
        self.old = debug_get(self.key)
        check(lib().inr_debug_set(self.key, self.value), "inr_debug_set")
        return self

    def __exit__(self, *exc):
        check(lib().inr_debug_set(self.key, self.old), "inr_debug_set")
        return False


# ---- measurement hooks ------------------------------------------------------------------------------------
def prof_enable(on: bool):
    check(lib().inr_prof_enable(1 if on else 0))


def prof_reset():
    check(lib().inr_prof_reset())


def prof_read(kernel_class: int):
    n = C.c_int64(0)
    ms = C.c_double(0.0)
    check(lib().inr_prof_read(int(kernel_class), C.byref(n), C.byref(ms)), "inr_prof_read")
    return int(n.value), float(ms.value)


def sincos_probe(x):
    _chk(x, "x")
    s = torch.empty_like(x)
    c = torch.empty_like(x)
    check(lib().inr_sincos_probe(s.data_ptr(), c.data_ptr(), x.data_ptr(), x.numel(), _stream()))
    return s, c
