"""Host-side mirror of the reference's INR module surface (SRDWI.py / INRmodel.py / nn_mri.py) on
top of the HIP kernels in libinrhip.so.

Same names, argument meaning and error behaviour as the reference so that its drivers
(superresDWI.py, superresHybrid.py, master.py, inrDWI.py) can import from here unchanged:
``get_mgrid, ImageFitting_set, input_mapping, SineLayer, Siren, PN, calculate_ADC, resize_array,
calculate_combinations``.  Everything numerical on the fit path is executed by hand-written
gfx950 kernels through the C ABI; tensors therefore live on the HIP device (``.cuda()`` on them is a
no-op) and every call raises ``InrDeviceError`` / ``InrHipUnavailable`` instead of falling back when
no GPU or no built library is present.

On top of the mirrored surface this module adds the fused fast path the reference does not have:
``SirenFitter`` / ``fit_siren`` (the whole superresDWI.py:132-138 loop enqueued by one C call, no
per-step Python or host sync) and ``reconstruct`` (dense re-sampling without materialising the
[N_test, 2m] feature matrix, superresDWI.py:125-126,161).

Citations are relative to /root/reference/implicit-neural-representations.
"""
from __future__ import annotations

import itertools
import math

import numpy as np
import torch
from torch import nn

from . import ops
from .ops import InrDeviceError  # noqa: F401  (re-exported)


# ---------------------------------------------------------------------------------------------------
# a-1 grids, a-2 dataset, a-3 Fourier features
# ---------------------------------------------------------------------------------------------------
def get_mgrid(shape, dim=None):
    """Flattened [-1,1] coordinate grid, bit-exact with the reference.

    ``get_mgrid(shape_tuple)`` is SRDWI.py:12-18 / INRmodel.py:12-18; ``get_mgrid(sidelen, dim=2)`` is
    the 2-D form of nn_mri.py:87-94.  Returns a float32 device tensor ``[prod(shape), len(shape)]``.
    """
    if isinstance(shape, (int, np.integer)):
        shape = (int(shape),) * int(2 if dim is None else dim)
    elif dim is not None:
        raise TypeError("get_mgrid(shape_tuple) takes no dim; use get_mgrid(sidelen, dim)")
    return ops.mgrid(tuple(int(s) for s in shape))


def input_mapping(x, B):
    """SRDWI.py:111-116: identity when ``B is None``, else ``[sin(2*pi*x@B.T) | cos(2*pi*x@B.T)]``."""
    if B is None:
        return x
    if x.requires_grad:
        return _FourierFn.apply(x, B)
    return ops.fourier_map(x.contiguous(), B.contiguous())


class _FourierFn(torch.autograd.Function):
    """Differentiable input_mapping for the INRmodel flavour (coords not detached, INRmodel.py:147):
    d/dx [sin p | cos p] with p = 2*pi*x@B.T  ->  gx = 2*pi * ((g_s*cos p - g_c*sin p) @ B)."""

    @staticmethod
    def forward(ctx, x, B):
        out = ops.fourier_map(x.contiguous(), B.contiguous())
        ctx.save_for_backward(out, B)
        return out

    @staticmethod
    def backward(ctx, g):
        out, B = ctx.saved_tensors
        m = B.shape[0]
        g = g.contiguous()
        # dp = g_sin*cos(p) - g_cos*sin(p): two element-wise products on the HIP path, then dp @ (2*pi*B)
        dp = ops.mul(g[:, :m].contiguous(), out[:, m:].contiguous()) - ops.mul(g[:, m:].contiguous(),
                                                                               out[:, :m].contiguous())
        gx = ops.sine_layer_backward_input(dp, (2.0 * math.pi * B).contiguous(), None)
        return gx, None


class ImageFitting_set(torch.utils.data.Dataset):
    """SRDWI.py:20-39 (list of equally-shaped N-D ndarrays) and nn_mri.py:182-203 (list of square
    PIL-like images: objects with ``.size``).  ``pixels [K,N,1]`` and ``coords [K,N,d]`` are device
    tensors; ``__getitem__`` returns the whole tensors like the reference does."""

    def __init__(self, img_dataset):
        super().__init__()
        first = img_dataset[0]
        pil_like = not isinstance(first, np.ndarray) and hasattr(first, "size") and not hasattr(first, "shape")
        if pil_like:  # nn_mri flavour: ToTensor then Normalize(0.5, 0.5): pixel -> 2*pixel - 1 (nn_mri.py:174-180)
            # torchvision's ToTensor divides 8-bit images (PIL mode 'L' -> uint8) by 255 and leaves every other mode as it
            # is ('F' float32 -- what master.py:122 passes -- and 'I' int32 are only cast)
            raw = [np.array(im) for im in img_dataset]
            side = raw[0].shape[0]
            if any(a.shape != (side, side) for a in raw):
                raise ValueError("nn_mri-style ImageFitting_set needs square images of equal size")
            self.orig = np.stack([a.astype(np.float64) for a in raw])        # nn_mri.py:192: np.array(img) as it is
            self.mean = sum(self.orig) / len(self.orig)
            arrays = [a.astype(np.float32) / 255.0 if a.dtype == np.uint8 else a for a in raw]
            self.shape = tuple(first.size)
            flat = [(2.0 * torch.from_numpy(a).float() - 1.0).reshape(-1, 1) for a in arrays]
            grid_shape = (side, side)
        else:
            shape = tuple(first.shape)
            if any(tuple(im.shape) != shape for im in img_dataset):
                raise ValueError("all images must share one shape")
            self.shape = shape
            flat = [torch.from_numpy(np.ascontiguousarray(im)).float().reshape(-1, 1) for im in img_dataset]
            grid_shape = shape
        dev = ops.require_gpu()
        self.pixels = torch.stack(flat).to(dev)
        grid = ops.mgrid(grid_shape)
        self.coords = grid.unsqueeze(0).expand(len(flat), -1, -1).contiguous() if len(flat) > 1 else grid.unsqueeze(0)

    def __len__(self):
        return len(self.pixels)

    def __getitem__(self, idx):
        return self.coords, self.pixels


# ---------------------------------------------------------------------------------------------------
# a-4 / a-5  SineLayer, Siren
# ---------------------------------------------------------------------------------------------------
HP_AUTOGRAD = True      # Siren.forward under autograd: the fused fit's kernels where they serve the shape (_SirenHpFn)


class _SineLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, omega):
        stash = any(ctx.needs_input_grad[:3])
        x = x.contiguous()
        act, dact = ops.sine_layer_forward(x, weight.contiguous(), bias, omega, stash)
        if stash:
            ctx.save_for_backward(x, weight, dact)
        return act

    @staticmethod
    def backward(ctx, g):
        x, weight, dact = ctx.saved_tensors
        dz = ops.mul(g.contiguous(), dact)
        gx = gW = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            gW, gb = ops.linear_param_grad(dz, x)
        if ctx.needs_input_grad[0]:
            gx = ops.sine_layer_backward_input(dz, weight.contiguous(), None)
        return gx, gW, gb, None


class SineLayer(nn.Module):
    """SRDWI.py:41-64 / nn_mri.py:96-120: ``sin(omega_0 * (x W^T + b))``; first layer weights
    U(-1/in, 1/in), others U(+-sqrt(6/in)/omega_0); bias keeps nn.Linear's default init."""

    def __init__(self, in_features, out_features, bias=True, is_first=False, omega_0=30):
        super().__init__()
        self.omega_0 = omega_0
        self.is_first = is_first
        self.in_features = in_features
        self.linear = nn.Linear(in_features, out_features, bias=bias)  # host RNG draw, as the reference
        self.init_weights()

    def init_weights(self):
        bound = 1 / self.in_features if self.is_first else np.sqrt(6 / self.in_features) / self.omega_0
        with torch.no_grad():
            self.linear.weight.uniform_(-bound, bound)

    def forward(self, input):
        lead = input.shape[:-1]
        out = _SineLayerFn.apply(input.reshape(-1, input.shape[-1]), self.linear.weight, self.linear.bias,
                                 float(self.omega_0))
        return out.reshape(*lead, out.shape[-1])


class _SirenFn(torch.autograd.Function):
    """Whole-network forward/backward on the HIP kernels (the same launch sequence as inr_siren_fit):
    per-layer fused GEMM+sin with the omega*cos stash, head row-reduction, then backward through
    head -> layers with the activation derivative fused into the input-grad GEMM epilogue."""

    @staticmethod
    def forward(ctx, x, first_omega, hidden_omega, *params):
        n_sine = len(params) // 2 - 1
        stash = any(ctx.needs_input_grad)
        x = x.contiguous()
        acts, dacts = [x], []
        for l in range(n_sine):
            act, dact = ops.sine_layer_forward(acts[-1], params[2 * l].contiguous(), params[2 * l + 1],
                                               first_omega if l == 0 else hidden_omega, stash)
            acts.append(act)
            dacts.append(dact)
        y = ops.linear_head_forward(acts[-1], params[2 * n_sine].contiguous(), params[2 * n_sine + 1])
        if stash:
            ctx.n_sine = n_sine
            ctx.save_for_backward(*acts, *dacts, *[params[2 * l] for l in range(n_sine + 1)])
        return y

    @staticmethod
    def backward(ctx, gy):
        n = ctx.n_sine
        saved = ctx.saved_tensors
        acts, dacts, weights = saved[:n + 1], saved[n + 1:2 * n + 1], saved[2 * n + 1:]
        need_params = any(ctx.needs_input_grad[3:])
        grads = [None] * (2 * (n + 1))
        # bias gradients ride along with the passes that produce dz (head pass / input-grad GEMM epilogue)
        dz, gWh, gbh, gb_l = ops.linear_head_backward(gy.contiguous(), acts[n], dacts[n - 1], weights[n].contiguous(),
                                                      need_dz=True, need_param=need_params,
                                                      need_bias_last=need_params)
        grads[2 * n], grads[2 * n + 1] = gWh, gbh
        gx = None
        for l in range(n - 1, -1, -1):
            if need_params:
                grads[2 * l], _ = ops.linear_param_grad(dz, acts[l], need_bias=False)
                grads[2 * l + 1] = gb_l
            if l > 0:
                if need_params:
                    dz, gb_l = ops.sine_layer_backward_input(dz, weights[l].contiguous(), dacts[l - 1], need_bias=True)
                else:
                    dz = ops.sine_layer_backward_input(dz, weights[l].contiguous(), dacts[l - 1])
            elif ctx.needs_input_grad[0]:
                gx = ops.sine_layer_backward_input(dz, weights[0].contiguous(), None)
        return (gx, None, None, *grads)


class _TrainState:
    """What the autograd path on the fused fit's kernels keeps per model (``_SirenHpFn``): the flat parameter buffer the module's
    parameters are views of (network order, as ``inr_siren_param_offsets`` lays it out -- the kernels prepare all weight images
    from it in two launches) and a stash workspace that is handed from the forward to its backward."""

    def __init__(self, model):
        self.model = model
        self.desc = model.desc()
        self.total, self.offsets = ops.siren_param_layout(self.desc)
        self.flat = None
        self._views = []
        self._free_ws = None          # a workspace no pending forward owns
        self._last = None             # ((workspace data_ptr, x data_ptr, x._version, n, strides), x) of the last forward: x kept alive

    def ensure(self):
        params = self.model.layer_parameters()
        if self.flat is not None and self.flat.device == params[0].device and \
                all(p.data_ptr() == v.data_ptr() for p, v in zip(params, self._views)):
            return
        flat = torch.zeros(self.total, dtype=torch.float32, device=params[0].device)
        views = []
        for l, (w_off, b_off) in enumerate(self.offsets):
            w, b = params[2 * l], params[2 * l + 1]
            vw = flat[w_off:w_off + w.numel()].view_as(w)
            vb = flat[b_off:b_off + b.numel()].view_as(b)
            vw.copy_(w.detach())
            vb.copy_(b.detach())
            w.data, b.data = vw, vb
            views += [vw, vb]
        self.flat, self._views, self._last = flat, views, None

    def take_workspace(self, n, device):
        need = ops.siren_fit_workspace_bytes(self.desc, n)
        ws, self._free_ws = self._free_ws, None
        if ws is None or ws.numel() < need or ws.device != device:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            # A NEW allocation may have been handed the address of a workspace a dropped graph freed (a forward without backward
            # never gives its workspace back): the address alone would then match `_last` -- and the C side's stamp for it -- while
            # other tensors have owned the memory in between.  The operand image is only trusted on a workspace this state kept.
            self._last = None
        return ws

    def give_back(self, ws):
        if self._free_ws is None or ws.numel() >= self._free_ws.numel():
            self._free_ws = ws

    def split_grads(self, flat_grads):
        out = []
        for l, (w_off, b_off) in enumerate(self.offsets):
            w, b = self._views[2 * l], self._views[2 * l + 1]
            out += [flat_grads[w_off:w_off + w.numel()].view_as(w), flat_grads[b_off:b_off + b.numel()].view_as(b)]
        return out


class _SirenHpFn(torch.autograd.Function):
    """The reference's own loop -- ``out = INR(x)``, torch forms the loss, ``loss.backward()``, ``torch.optim.Adam.step()``
    (superresDWI.py:132-138) -- on the kernels of the fused fit: ONE call enqueues the weight preparation and the forward of every
    layer with its stash (``inr_siren_forward_train``), one call the whole backward pass and the fixed-order gradient reduction
    (``inr_siren_backward_train``).  ``_SirenFn`` below, the layer-by-layer form on the exact-fp32 kernels, remains for the shapes
    these kernels do not serve and for inputs that want their own gradient."""

    @staticmethod
    def forward(ctx, x, state, *params):
        x = x.contiguous()
        n = x.shape[0]
        ws = state.take_workspace(n, x.device)
        # The operand image of x is reused when the same memory, unmodified, comes again.  A data pointer and a version counter alone
        # do not say that -- a batch tensor freed after one step hands its address (and a version counter of 0) to the next step's
        # batch -- so the state KEEPS the last input alive (one tensor view; `Siren.forward` makes a new view object per call, so the
        # Python object's identity says nothing either): while it lives, nothing else can own that address.
        key = (ws.data_ptr(), x.data_ptr(), x._version, n, tuple(x.stride()))
        same = state._last is not None and state._last[0] == key
        y, ws = ops.siren_forward_train(state.desc, state.flat, x, ws, ops.REUSE_INPUT_IMAGE if same else 0)
        state._last = (key, x)
        ctx.state, ctx.ws, ctx.x = state, ws, x          # (x kept alive: the backward's layer-0 GEMM reads its operand image only,
        return y                                         #  but the C side remembers the pointer)

    @staticmethod
    def backward(ctx, gy):
        st = ctx.state
        if ctx.ws is None:
            raise RuntimeError("the fused SIREN path runs ONE backward per forward (its stash workspace went back to the pool after the "
                               "first); for retain_graph=True or two losses sharing one forward set "
                               "mri_super_resolution_amd.inr.HP_AUTOGRAD = False (layer-by-layer exact-fp32 path)")
        grads = torch.empty(st.total, dtype=torch.float32, device=gy.device)      # fresh: .grad may alias what is returned here
        ops.siren_backward_train(st.desc, st.flat, grads, gy.contiguous(), ctx.ws)
        st.give_back(ctx.ws)
        ctx.ws = None
        return (None, None, *st.split_grads(grads))


class Siren(nn.Module):
    """``Siren(in_features, hidden_features, hidden_layers, out_features, first_omega_0=30.,
    hidden_omega_0=30.)`` -- SRDWI.py:67-91 (``flavor='SRDWI'``, also nn_mri.py:122-146) or
    INRmodel.py:122-151 (``flavor='INRmodel'``).

    The flavours differ exactly where the reference modules do: RNG draw order at construction
    (head first vs. last, so one seed gives the reference's weights) and whether ``forward`` detaches
    the coordinates.  ``state_dict()`` keys are the reference's: ``final_linear.*``,
    ``net.k.linear.*`` and the alias ``net.{L+1}.*``.  ``return_coords=True`` restores the original
    SIREN ``(output, coords)`` tuple that master.py:142 unpacks.
    Initialisation happens on the host with torch's CPU generator (bit-identical weights); the
    forward/backward only exists on a HIP device.
    """

    def __init__(self, in_features, hidden_features, hidden_layers, out_features, first_omega_0=30.,
                 hidden_omega_0=30., flavor="SRDWI", return_coords=False):
        super().__init__()
        if flavor not in ("SRDWI", "INRmodel"):
            raise ValueError("flavor must be 'SRDWI' or 'INRmodel'")
        self.flavor = flavor
        self.return_coords = return_coords
        self.in_features, self.hidden_features = int(in_features), int(hidden_features)
        self.hidden_layers, self.out_features = int(hidden_layers), int(out_features)
        self.first_omega_0, self.hidden_omega_0 = float(first_omega_0), float(hidden_omega_0)
        head_bound = np.sqrt(6 / hidden_features) / hidden_omega_0

        def head():
            lin = nn.Linear(hidden_features, out_features)
            with torch.no_grad():
                lin.weight.uniform_(-head_bound, head_bound)
            return lin

        if flavor == "SRDWI":
            self.final_linear = head()
        layers = [SineLayer(in_features, hidden_features, is_first=True, omega_0=first_omega_0)]
        for _ in range(hidden_layers):
            layers.append(SineLayer(hidden_features, hidden_features, is_first=False, omega_0=hidden_omega_0))
        if flavor == "INRmodel":
            self.final_linear = head()
        layers.append(self.final_linear)
        self.net = nn.Sequential(*layers)

    # -- parameter access in network order (W_0, b_0, ..., W_head, b_head) -------------------------------
    def layer_parameters(self):
        out = []
        for mod in self.net:
            lin = mod.linear if isinstance(mod, SineLayer) else mod
            out += [lin.weight, lin.bias]
        return out

    def desc(self):
        return ops.make_desc(self.in_features, self.hidden_features, self.hidden_layers, self.out_features,
                             self.first_omega_0, self.hidden_omega_0)

    def forward(self, coords):
        if self.flavor == "SRDWI":
            coords = coords.detach()  # SRDWI.py:88 (the reference also clones; the kernels never write x)
        lead = coords.shape[:-1]
        flat_x = coords.reshape(-1, coords.shape[-1])
        params = self.layer_parameters()
        if self._hp_train_ok(flat_x, params):
            st = self.__dict__.get("_hp_state")
            if st is None:
                st = self.__dict__["_hp_state"] = _TrainState(self)
            st.ensure()
            y = _SirenHpFn.apply(flat_x, st, *self.layer_parameters())
        else:
            y = _SirenFn.apply(flat_x, self.first_omega_0, self.hidden_omega_0, *params)
        y = y.reshape(*lead, self.out_features)
        return (y, coords) if self.return_coords else y

    def _hp_train_ok(self, x, params) -> bool:
        """The fused kernels take a training forward when every parameter wants a gradient, the input does not, everything is
        fp32 on the device and the shape is one they serve; ``inr.HP_AUTOGRAD = False`` keeps the layer-by-layer path (A/B)."""
        if not (HP_AUTOGRAD and torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and not x.requires_grad):
            return False
        if not all(p.requires_grad and p.is_cuda and p.dtype == torch.float32 for p in params):
            return False
        # (asked every time: eligibility also depends on the process-global diagnostic switches -- a host-only call, no device work)
        return ops.siren_hp_eligible(self.desc())


# ---------------------------------------------------------------------------------------------------
# a-8 fused fit loop, a-9 dense re-sampling
# ---------------------------------------------------------------------------------------------------
class SirenFitter:
    """Fused replacement of the reference's per-step Python loop (superresDWI.py:132-138,
    superresHybrid.py:109-114; weighted form master.py:143-148): forward, MSE, backward and Adam of
    ``n_steps`` full-batch steps are enqueued by ONE call into ``inr_siren_fit``.

    The model's parameters are re-pointed at views of one flat fp32 buffer (network order, padded to
    16 B per tensor), so ``model.state_dict()`` / an external optimizer keep seeing live weights.
    Adam state (``m``, ``v``, step count) lives here, which is what lets a fit be continued or
    interleaved with other phases (superresDWI.py:139-156).
    """

    def __init__(self, model: Siren, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        ops.require_gpu()
        self.model = model
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.desc = model.desc()
        self.total, self.offsets = ops.siren_param_layout(self.desc)
        self.step_count = 0
        self.flat = None
        self._workspace = None
        self._adopt()

    def _adopt(self):
        params = self.model.layer_parameters()
        dev = params[0].device
        if not params[0].is_cuda:
            raise InrDeviceError("move the model to the HIP device first (model.cuda())")
        flat = torch.zeros(self.total, dtype=torch.float32, device=dev)
        views = []
        for l, (w_off, b_off) in enumerate(self.offsets):
            w, b = params[2 * l], params[2 * l + 1]
            vw = flat[w_off:w_off + w.numel()].view_as(w)
            vb = flat[b_off:b_off + b.numel()].view_as(b)
            vw.copy_(w.detach())
            vb.copy_(b.detach())
            w.data, b.data = vw, vb
            views += [vw, vb]
        old = self.flat
        self.flat = flat
        self._views = views
        if old is None:
            self.grads = torch.zeros_like(flat)
            self.m = torch.zeros_like(flat)
            self.v = torch.zeros_like(flat)

    def _check_views(self):
        params = self.model.layer_parameters()
        if any(p.data_ptr() != v.data_ptr() for p, v in zip(params, self._views)):
            self._adopt()  # model was moved / reloaded: re-flatten (Adam state is kept)

    def step(self, model_input, target, n_steps=1, weight=None):
        """Run ``n_steps`` fit steps; returns the per-step losses as a device tensor (no sync)."""
        self._check_views()
        x = model_input.detach().reshape(-1, model_input.shape[-1]).contiguous()
        t = target.detach().reshape(-1).contiguous()
        w = None if weight is None else weight.detach().reshape(-1).contiguous()
        losses = torch.empty(max(int(n_steps), 1), dtype=torch.float32, device=x.device)
        need = ops.siren_fit_workspace_bytes(self.desc, x.shape[0])
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        ops.siren_fit(self.desc, self.flat, self.grads, self.m, self.v, x, t, w, self.step_count + 1, int(n_steps),
                      self.lr, self.betas[0], self.betas[1], self.eps, losses, self._workspace)
        self.step_count += int(n_steps)
        return losses[:n_steps]

    def step_cycle(self, model_input, targets, n_steps, weights=None, first_acq=0):
        """``n_steps`` steps whose target (and weight) image changes every step (master.py:137-148): ``targets`` /
        ``weights`` are [n_acq, N] device tensors, step ``it`` fits acquisition ``(first_acq + it) % n_acq``.  One call into
        ``inr_siren_fit_cycle`` -- for the small master.py networks one persistent launch per 64 steps."""
        self._check_views()
        x = model_input.detach().reshape(-1, model_input.shape[-1]).contiguous()
        t = targets.detach().reshape(targets.shape[0], -1).contiguous()
        w = None if weights is None else weights.detach().reshape(weights.shape[0], -1).contiguous()
        losses = torch.empty(max(int(n_steps), 1), dtype=torch.float32, device=x.device)
        need = ops.siren_fit_workspace_bytes(self.desc, x.shape[0])
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=x.device)
        ops.siren_fit_cycle(self.desc, self.flat, self.grads, self.m, self.v, x, t, w, int(first_acq), self.step_count + 1,
                            int(n_steps), self.lr, self.betas[0], self.betas[1], self.eps, losses, self._workspace)
        self.step_count += int(n_steps)
        return losses[:n_steps]

    def release_workspace(self):
        self._workspace = None


class ShardedSirenFitter(SirenFitter):
    """One fit whose coordinate rows are split over the ranks of a process group (SURVEY.md 8 e): every rank holds
    identical weights and its own row shard; per step the local forward/backward (``inr_siren_loss_grad``, mean taken
    over the GLOBAL row count) is followed by ONE all-reduce(sum) of the flat gradient with the loss riding in a spare
    slot behind it -- 3.68 MB for Siren(256,512,3,1) over RCCL/xGMI -- and an identical local Adam step.  Mathematically
    the full-batch step of superresDWI.py:134-138; the summation order differs from the single-GPU run (tier T3/T4 parity,
    not bitwise).  At construction the weights and the Adam state of the group's first rank are broadcast, so ranks
    that drew different initialisations (unseeded RNG) still fit ONE network."""

    def __init__(self, model: Siren, global_rows: int, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, group=None):
        super().__init__(model, lr=lr, betas=betas, eps=eps)
        self.global_rows = int(global_rows)
        self.group = group
        # gradient buffer with the loss slot behind it: one collective per step
        self._gbuf = torch.zeros(self.total + 4, dtype=torch.float32, device=self.flat.device)
        self.grads = self._gbuf[:self.total]
        self._loss = self._gbuf[self.total:self.total + 1]
        self.sync_replicas()

    def _world(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def _all_reduce(self, t):
        import torch.distributed as dist
        from .dist import is_shared
        if not is_shared(self._world()):
            return
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:   # gloo (CPU tests): stage through host memory
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)

    def _broadcast(self, t):
        import torch.distributed as dist
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        if dist.get_backend(self.group) == "nccl":
            dist.broadcast(t, src=src, group=self.group)
        else:
            h = t.cpu()
            dist.broadcast(h, src=src, group=self.group)
            t.copy_(h)

    def sync_replicas(self):
        """Every rank takes the weights, Adam moments and step count of the group's first rank."""
        from .dist import is_shared
        if not is_shared(self._world()):
            return
        count = torch.tensor([float(self.step_count)], dtype=torch.float64, device=self.flat.device)
        for t in (self.flat, self.m, self.v, count):
            self._broadcast(t)
        self.step_count = int(count.item())

    def step(self, model_input, target, n_steps=1, weight=None):
        self._check_views()
        x = model_input.detach().reshape(-1, model_input.shape[-1]).contiguous()
        t = target.detach().reshape(-1).contiguous()
        w = None if weight is None else weight.detach().reshape(-1).contiguous()
        losses = torch.empty(max(int(n_steps), 1), dtype=torch.float32, device=x.device)
        count_total = self.global_rows * self.desc.out_features
        for it in range(int(n_steps)):
            # from the second step of this call on, x / target / weight are unchanged: keep their operand image and maxima
            flags = (ops.REUSE_INPUT_IMAGE | ops.REUSE_TARGET_STATS) if (it > 0 and self._workspace is not None) else 0
            self._workspace = ops.siren_loss_grad(self.desc, self.flat, self.grads, x, t, w, count_total, self._loss,
                                                  self._workspace, flags)
            self._all_reduce(self._gbuf)          # gradient + loss in one message
            self.step_count += 1
            ops.adam_step(self.flat, self.grads, self.m, self.v, self.step_count, self.lr, self.betas[0], self.betas[1],
                          self.eps)
            losses[it:it + 1].copy_(self._loss)   # device-side, no host sync
        return losses[:n_steps]

    def step_cycle(self, model_input, targets, n_steps, weights=None, first_acq=0):
        """The cycling-acquisition loop (master.py:137-148) on a row-sharded fit: ``targets`` / ``weights`` are this rank's
        [n_acq, local rows]; every step is local forward/backward on acquisition ``(first_acq + it) % n_acq``, the one
        all-reduce, the identical Adam step.  (The base class would run the fused kernel on the local shard alone -- no
        all-reduce, local row count in the mean -- and the replicas would drift apart.)"""
        self._check_views()
        x = model_input.detach().reshape(-1, model_input.shape[-1]).contiguous()
        t = targets.detach().reshape(targets.shape[0], -1).contiguous()
        w = None if weights is None else weights.detach().reshape(weights.shape[0], -1).contiguous()
        if w is not None and tuple(w.shape) != tuple(t.shape):
            raise ValueError("weights must have the shape of targets")
        n_acq = t.shape[0]
        if not 0 <= int(first_acq) < n_acq:
            raise ValueError("first_acq out of range")
        losses = torch.empty(max(int(n_steps), 1), dtype=torch.float32, device=x.device)
        count_total = self.global_rows * self.desc.out_features
        for it in range(int(n_steps)):
            a = (int(first_acq) + it) % n_acq
            flags = ops.REUSE_INPUT_IMAGE if (it > 0 and self._workspace is not None) else 0      # (the targets do change)
            self._workspace = ops.siren_loss_grad(self.desc, self.flat, self.grads, x, t[a], None if w is None else w[a],
                                                  count_total, self._loss, self._workspace, flags)
            self._all_reduce(self._gbuf)
            self.step_count += 1
            ops.adam_step(self.flat, self.grads, self.m, self.v, self.step_count, self.lr, self.betas[0], self.betas[1],
                          self.eps)
            losses[it:it + 1].copy_(self._loss)
        return losses[:n_steps]


def flat_parameters(model: Siren):
    """Flat fp32 parameter buffer in the C ABI's layout (a copy; for inference entry points)."""
    desc = model.desc()
    total, offsets = ops.siren_param_layout(desc)
    params = model.layer_parameters()
    if not params[0].is_cuda:
        raise InrDeviceError("move the model to the HIP device first (model.cuda())")
    flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)
    for l, (w_off, b_off) in enumerate(offsets):
        flat[w_off:w_off + params[2 * l].numel()] = params[2 * l].detach().reshape(-1)
        flat[b_off:b_off + params[2 * l + 1].numel()] = params[2 * l + 1].detach().reshape(-1)
    return desc, flat


def fit_siren(model: Siren, model_input, target, steps, lr=1e-4, weight=None, chunk=250, fitter=None):
    """Convenience wrapper: a full fit in chunks of ``chunk`` steps.  Returns (fitter, losses tensor)."""
    fitter = fitter or SirenFitter(model, lr=lr)
    out = []
    done = 0
    while done < steps:
        k = min(chunk, steps - done)
        out.append(fitter.step(model_input, target, k, weight))
        done += k
    return fitter, (torch.cat(out) if out else torch.empty(0))


def reconstruct(model: Siren, shape, B=None, clamp_min=0.0, chunk_rows=1 << 20):
    """``clamp(INR(input_mapping(get_mgrid(shape), B)), min=0).view(shape)`` (superresDWI.py:125-126,
    161-162; superresHybrid.py:103-104,119) as one C call: grid and Fourier features are produced
    chunk by chunk on the device.  ``clamp_min=None`` skips the clamp (master.py:149-153)."""
    desc, flat = flat_parameters(model)
    Bd = None if B is None else B.detach().to(flat.device, torch.float32).contiguous()
    y = ops.siren_reconstruct(desc, flat, shape, Bd, clamp_min, chunk_rows)
    shape = tuple(int(s) for s in shape)
    return y.view(*shape) if model.out_features == 1 else y.view(*shape, model.out_features)


# ---------------------------------------------------------------------------------------------------
# a-10 PerturbNet
# ---------------------------------------------------------------------------------------------------
class _PNFn(torch.autograd.Function):
    """PerturbNet forward/backward on the HIP kernels: tanh hidden layer = the fp32-MFMA GEMM with a tanh
    epilogue, output layer = shuffle-reduced row dots with eps*tanh; the constant acquisition column
    (SRDWI.py:102-104) is folded into the hidden bias instead of concatenating an [N, F+1] matrix."""

    @staticmethod
    def forward(ctx, x, acq, eps, w1, b1, w2, b2):
        stash = any(ctx.needs_input_grad[3:])
        fin = x.shape[1]
        w1a = w1[:, :fin].contiguous()
        b_eff = (b1 + acq * w1[:, fin]).contiguous()
        hid, dhid = ops.tanh_layer_forward(x.contiguous(), w1a, b_eff, 1.0, stash)
        out, dout = ops.linear_tanh_head_forward(hid, w2.contiguous(), b2, eps, stash)
        if stash:
            ctx.acq = acq
            ctx.save_for_backward(x, hid, dhid, dout, w2)
        return out

    @staticmethod
    def backward(ctx, g):
        x, hid, dhid, dout, w2 = ctx.saved_tensors
        dz2 = ops.mul(g.contiguous(), dout)
        dz1, gw2, gb2, gb1 = ops.linear_head_backward(dz2, hid, dhid, w2.contiguous(), need_dz=True, need_param=True,
                                                      need_bias_last=True)
        gw1a, _ = ops.linear_param_grad(dz1, x, need_bias=False)
        gw1 = torch.cat([gw1a, (ctx.acq * gb1).unsqueeze(1)], dim=1)
        return None, None, None, gw1, gb1, gw2, gb2


class PN(nn.Module):
    """SRDWI.py:93-109 (``dimension`` outputs) / nn_mri.py:148-164 (2 outputs when ``dimension`` is omitted):
    ``eps * tanh(Linear2(tanh(Linear1(cat(coords, sample/10)))))`` with the input detached (SRDWI.py:101)."""

    def __init__(self, in_features, hidden_features, dimension=2):
        super().__init__()
        self.tanh = nn.Tanh()
        self.perturb_linear = nn.Linear(in_features + 1, hidden_features)
        self.perturb_linear2 = nn.Linear(hidden_features, dimension)

    def forward(self, coords, sample=0, eps=0):
        coords = coords.detach()
        acq = float(np.float32(sample / 10.))      # torch.tensor([sample/10.], dtype=torch.float) in the reference
        return _PNFn.apply(coords.reshape(-1, coords.shape[-1]), acq, float(eps), self.perturb_linear.weight,
                           self.perturb_linear.bias, self.perturb_linear2.weight, self.perturb_linear2.bias)


# ---------------------------------------------------------------------------------------------------
# host-side helpers of the reference surface (not on the device path)
# ---------------------------------------------------------------------------------------------------
def calculate_ADC(bvalues, slicedata):
    """SRDWI.py:118-130 -- the per-pixel ``np.polyfit(b/1000, log(S+1e-7), 1)`` slope, in closed form
    over the whole slice at once; ADC = -slope clipped to [-10, 3]."""
    b = np.asarray(bvalues, np.float64).reshape(-1) / 1000.0
    logs = np.log(np.asarray(slicedata, np.float64) + 1e-7)
    bc = b - b.mean()
    slope = (bc * (logs - logs.mean(axis=-1, keepdims=True))).sum(-1) / (bc * bc).sum()
    return np.clip(-slope, -10.0, 3.0)


def resize_array(arr, new_size=128, kind='cubic'):
    """SRDWI.py:132-141: spline interpolation of the third axis to ``new_size`` samples."""
    from scipy.interpolate import interp1d
    old = arr.shape[2]
    f = interp1d(np.linspace(0, 1, old), arr, kind=kind, axis=2)
    return np.asarray(f(np.linspace(0, 1, new_size)), np.float64)


def calculate_combinations(voxel, hybrid_raw_norm):
    """SRDWI.py:143-152: all products of one acquisition per b-value (TE index 0) at one voxel."""
    i, j, k = voxel
    te = 0
    per_b = [[hybrid_raw_norm[0][te][i, j, k]]]
    for b in (1, 2, 3):
        per_b.append(list(hybrid_raw_norm[b][te][i, j, k, :]))
    return np.asarray(list(itertools.product(*per_b))).T
