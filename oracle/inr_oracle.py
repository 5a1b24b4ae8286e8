"""CPU oracle for the INR (SIREN) super-resolution hot path -- TEST INFRASTRUCTURE ONLY.

This file is a plain NumPy restatement of the arithmetic the reference performs with stock
PyTorch ops on the path named in SURVEY.md section 8.  It is the *checker* for the HIP
kernels: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product package (``mri-super-resolution_amd``) never imports it and
fails loudly when ``libinrhip.so`` or a GPU is missing.

Parity status: PINNED.  Every function here is checked in ``tests/test_oracle_golden.py``
against fixtures under ``tests/golden/`` that were produced by importing the reference's
own ``SRDWI.py`` / ``INRmodel.py`` in the build container (generator: ``oracle/gen_golden.py``).

All ``file:line`` citations are relative to ``/root/reference/implicit-neural-representations``.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
TWO_PI_F32 = F32(2.0 * np.pi)  # torch multiplies an fp32 tensor by the python scalar 2*pi -> fp32 scalar


# --------------------------------------------------------------------------------------
# a-1  coordinate grids                                     SRDWI.py:12-18, nn_mri.py:87-94
# --------------------------------------------------------------------------------------
def linspace_pm1(n: int) -> np.ndarray:
    """``torch.linspace(-1, 1, steps=n)`` in fp32, bit for bit (SRDWI.py:15).

    Rule (verified for n = 1..700 and large n against torch 2.10 CPU): step = 2f/(n-1)f;
    lower half  v[i] = fma(step, i, -1); upper half v[i] = fma(-step, n-1-i, 1); each a single
    rounding.  The float64 expression below is exact before the final cast for every n < 2**20
    (24-bit step times < 2**24 integer fits in 53 bits; the sum with +-1 needs < 53 bits).
    """
    if n <= 0:
        return np.empty((0,), F32)
    if n == 1:
        return np.array([-1.0], F32)
    step = F32(2.0) / F32(n - 1)
    i = np.arange(n, dtype=np.float64)
    lower = (np.float64(step) * i - 1.0).astype(F32)
    upper = (1.0 - np.float64(step) * (n - 1 - i)).astype(F32)
    return np.where(np.arange(n) < n // 2, lower, upper).astype(F32)


def mgrid(shape) -> np.ndarray:
    """``get_mgrid(shape)`` (SRDWI.py:12-18): meshgrid 'ij', stacked on the last axis, flattened
    row-major (last axis fastest).  Returns fp32 ``[prod(shape), len(shape)]``."""
    shape = tuple(int(s) for s in shape)
    axes = [linspace_pm1(n) for n in shape]
    d = len(shape)
    n = int(np.prod(shape)) if d else 0
    out = np.empty((n, d), F32)
    for a, ax in enumerate(axes):
        view = [1] * d
        view[a] = shape[a]
        out[:, a] = np.broadcast_to(ax.reshape(view), shape).reshape(-1)
    return out


def mgrid_square(sidelen: int, dim: int = 2) -> np.ndarray:
    """2-D flavour ``get_mgrid(sidelen, dim)`` (nn_mri.py:87-94)."""
    return mgrid((int(sidelen),) * int(dim))


# --------------------------------------------------------------------------------------
# a-2  dataset flattening                                            SRDWI.py:20-39
# --------------------------------------------------------------------------------------
def image_fitting_set(images):
    """``ImageFitting_set`` (SRDWI.py:24-33): pixels[k] = C-order flatten of image k cast to fp32
    as a column; coords[k] = mgrid(shape) (the same grid repeated per image)."""
    shape = tuple(images[0].shape)
    n = int(np.prod(shape))
    pixels = np.empty((len(images), n, 1), F32)
    coords = np.empty((len(images), n, len(shape)), F32)
    grid = mgrid(shape)
    for k, img in enumerate(images):
        pixels[k, :, 0] = np.ascontiguousarray(img).astype(F32).reshape(-1)
        coords[k] = grid
    return pixels, coords, shape


# --------------------------------------------------------------------------------------
# a-3  Fourier-feature input mapping                                 SRDWI.py:111-116
# --------------------------------------------------------------------------------------
def fourier_features(x: np.ndarray, B, dtype=F32) -> np.ndarray:
    """``input_mapping(x, B)``: ``[sin(2*pi*x @ B.T) | cos(2*pi*x @ B.T)]`` (sin block first).
    ``B is None`` is the identity (SRDWI.py:112-113).  With ``dtype=float32`` the scaling by the
    fp32-rounded 2*pi and the tiny-K contraction are done in fp32 like torch does."""
    if B is None:
        return x
    x = np.asarray(x, dtype)
    Bm = np.asarray(B, dtype)
    scaled = (dtype(TWO_PI_F32) * x).astype(dtype)
    proj = np.zeros((x.shape[0], Bm.shape[0]), dtype)
    for j in range(x.shape[1]):  # K = d is 2..4: explicit, fixed summation order
        proj = (proj + scaled[:, j:j + 1] * Bm[:, j][None, :]).astype(dtype)
    return np.concatenate([np.sin(proj), np.cos(proj)], axis=-1).astype(dtype)


# --------------------------------------------------------------------------------------
# a-4/a-5  SIREN forward                                             SRDWI.py:41-91
# --------------------------------------------------------------------------------------
def siren_forward(weights, biases, x, first_omega=30.0, hidden_omega=30.0, dtype=F32, stash=False):
    """``Siren.forward``: ``a_{l+1} = sin(omega_l * (a_l @ W_l.T + b_l))`` for the L+1 sine layers,
    then the linear head ``y = a @ W_head.T + b_head`` (SRDWI.py:58-59, 75-91).

    ``weights``/``biases`` are listed in network order (layer 0 .. L, head last).  With
    ``stash=True`` also returns the layer inputs ``a_l`` and pre-activations ``z_l`` for backward.
    """
    a = np.asarray(x, dtype)
    acts, pre = [a], []
    n_sine = len(weights) - 1
    for l in range(n_sine):
        om = dtype(first_omega if l == 0 else hidden_omega)
        z = (a @ np.asarray(weights[l], dtype).T + np.asarray(biases[l], dtype)).astype(dtype)
        a = np.sin(om * z).astype(dtype)
        pre.append(z)
        acts.append(a)
    y = (a @ np.asarray(weights[-1], dtype).T + np.asarray(biases[-1], dtype)).astype(dtype)
    if stash:
        return y, acts, pre
    return y


# --------------------------------------------------------------------------------------
# a-6  loss + analytic backward        superresDWI.py:135-137, master.py:143-147 (autograd there)
# --------------------------------------------------------------------------------------
def mse_loss_and_grad(y, target, weight=None, dtype=F32):
    """``L = mean(w * (y - t)**2)`` over all N*out elements; returns (L, dL/dy)."""
    y = np.asarray(y, dtype)
    t = np.asarray(target, dtype)
    r = y - t
    n = dtype(r.size)
    if weight is None:
        loss = dtype(np.mean(r.astype(np.float64) ** 2))
        g = (dtype(2.0) * r / n).astype(dtype)
    else:
        w = np.asarray(weight, dtype)
        loss = dtype(np.mean(w.astype(np.float64) * r.astype(np.float64) ** 2))
        g = (dtype(2.0) * w * r / n).astype(dtype)
    return loss, g


def siren_backward(weights, acts, pre, gy, first_omega=30.0, hidden_omega=30.0, dtype=F32):
    """Gradients of every weight/bias given ``gy = dL/dy``; restates what autograd produces for
    SRDWI.py:58-59: ``dz_l = da_{l+1} * omega*cos(omega*z_l)``, ``dW_l = dz_l.T @ a_l``,
    ``db_l = sum_rows dz_l``, ``da_l = dz_l @ W_l`` (layer 0 input needs no gradient,
    SRDWI.py:88).  Returns (grad_weights, grad_biases) in network order."""
    n_sine = len(weights) - 1
    gw = [None] * (n_sine + 1)
    gb = [None] * (n_sine + 1)
    gy = np.asarray(gy, dtype)
    gw[-1] = (gy.T @ acts[-1]).astype(dtype)
    gb[-1] = gy.sum(axis=0).astype(dtype)
    da = (gy @ np.asarray(weights[-1], dtype)).astype(dtype)
    for l in range(n_sine - 1, -1, -1):
        om = dtype(first_omega if l == 0 else hidden_omega)
        dz = (da * (om * np.cos(om * pre[l]))).astype(dtype)
        gw[l] = (dz.T @ acts[l]).astype(dtype)
        gb[l] = dz.sum(axis=0).astype(dtype)
        if l > 0:
            da = (dz @ np.asarray(weights[l], dtype)).astype(dtype)
    return gw, gb


# --------------------------------------------------------------------------------------
# a-7  Adam                       torch.optim.Adam defaults as used at superresDWI.py:116,138
# --------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """One Adam update following torch's single-tensor path: ``m.lerp_(g, 1-b1)``,
    ``v = b2*v + (1-b2)*g*g``; bias corrections and step size in double on the host;
    ``denom = sqrt(v)/sqrt(bc2) + eps``; ``p -= (lr/bc1) * m/denom``.  fp32 state, in place.
    ``step`` is the 1-based step count *after* increment."""
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    bc2_sqrt = bc2 ** 0.5
    m += (g - m) * F32(1.0 - beta1)
    v *= F32(beta2)
    v += F32(1.0 - beta2) * g * g
    denom = (np.sqrt(v) / F32(bc2_sqrt) + F32(eps)).astype(F32)
    p -= (F32(step_size) * (m / denom)).astype(F32)
    return p, m, v


# --------------------------------------------------------------------------------------
# a-9  dense re-sampling           superresDWI.py:125-126,161-162; superresHybrid.py:103-104,119
# --------------------------------------------------------------------------------------
def reconstruct(weights, biases, shape, B, clamp_min=0.0, first_omega=30.0, hidden_omega=30.0,
                dtype=F32, chunk=65536):
    """``clamp(INR(input_mapping(get_mgrid(shape), B)), min=0).view(shape)``."""
    grid = mgrid(shape)
    out = np.empty((grid.shape[0],), dtype)
    for s in range(0, grid.shape[0], chunk):
        feats = fourier_features(grid[s:s + chunk], B, dtype)
        y = siren_forward(weights, biases, feats, first_omega, hidden_omega, dtype)
        out[s:s + chunk] = y[:, 0]
    if clamp_min is not None:
        out = np.maximum(out, dtype(clamp_min))
    return out.reshape(shape)


# --------------------------------------------------------------------------------------
# a-10  PerturbNet forward                                           SRDWI.py:93-109
# --------------------------------------------------------------------------------------
def pn_forward(w1, b1, w2, b2, coords, sample=0, eps=0.0, dtype=F32):
    """``PN.forward``: concat(coords, sample/10) -> Linear -> tanh -> Linear -> eps*tanh."""
    c = np.asarray(coords, dtype)
    acq = np.full((c.shape[0], 1), dtype(np.float32(sample / 10.0)), dtype)
    h = np.tanh((np.concatenate([c, acq], -1) @ np.asarray(w1, dtype).T + np.asarray(b1, dtype)).astype(dtype))
    o = (h @ np.asarray(w2, dtype).T + np.asarray(b2, dtype)).astype(dtype)
    return (dtype(eps) * np.tanh(o)).astype(dtype)


# --------------------------------------------------------------------------------------
# a-12  metrics
# --------------------------------------------------------------------------------------
def psnr(ref, test, data_range=1.0):
    """``10*log10(data_range**2 / MSE)`` -- the definition of skimage's peak_signal_noise_ratio
    (imported but unused at master.py:14); float64 accumulation."""
    ref = np.asarray(ref, np.float64)
    test = np.asarray(test, np.float64)
    mse = np.mean((ref - test) ** 2)
    return float(10.0 * np.log10((data_range ** 2) / mse))


def _box_filter_2d(img, win):
    """Uniform filter with scipy.ndimage 'reflect' boundaries (what skimage 0.20 uses)."""
    from scipy.ndimage import uniform_filter
    return uniform_filter(img, size=win, mode="reflect")


def ssim2d(im1, im2, data_range=1.0, win=7, k1=0.01, k2=0.03):
    """skimage-0.20 ``structural_similarity`` defaults for 2-D float images as called at
    superresDWI.py:186: uniform 7x7 window, sample covariance (N/(N-1)), mean over the image
    cropped by (win-1)//2 on every side.  Parity with skimage itself is UNPINNED here (package
    absent); pinned by analytic cases in tests."""
    x = np.asarray(im1, np.float64)
    y = np.asarray(im2, np.float64)
    npix = win * win
    cov_norm = npix / (npix - 1.0)
    ux, uy = _box_filter_2d(x, win), _box_filter_2d(y, win)
    uxx, uyy, uxy = _box_filter_2d(x * x, win), _box_filter_2d(y * y, win), _box_filter_2d(x * y, win)
    vx = cov_norm * (uxx - ux * ux)
    vy = cov_norm * (uyy - uy * uy)
    vxy = cov_norm * (uxy - ux * uy)
    c1 = (k1 * data_range) ** 2
    c2 = (k2 * data_range) ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
    pad = (win - 1) // 2
    return float(s[pad:s.shape[0] - pad, pad:s.shape[1] - pad].mean())


def adc_map(bvalues, slicedata, min_adc=-10.0, max_adc=3.0, eps=1e-7):
    """``calculate_ADC`` (SRDWI.py:118-130): per pixel, minus the slope of the degree-1 least-squares
    fit of log(S + eps) against b/1000, clipped to [min_adc, max_adc].  Closed form of np.polyfit."""
    b = np.asarray(bvalues, np.float64).reshape(-1) / 1000.0
    y = np.log(np.asarray(slicedata, np.float64) + eps)
    bm = b.mean()
    slope = ((b - bm) * (y - y.mean(axis=-1, keepdims=True))).sum(-1) / ((b - bm) ** 2).sum()
    return np.clip(-slope, min_adc, max_adc)


def rel_l2(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


# ---- spline baseline (superresDWI.py:172-191: skimage.transform.rescale(img, s, anti_aliasing=True)) -----------------------
def rescale_linear(img, scale):
    """skimage 0.20 ``rescale`` for ``scale >= 1`` on a 2-D float image, restated: ``resize`` with order 1 and mode 'reflect'
    = ``scipy.ndimage.zoom(img, out/in, order=1, mode='mirror', grid_mode=True)`` after a sigma-0 (identity) anti-aliasing
    filter.  Pinned against that scipy call in tests/test_baselines_cpu.py (scikit-image itself is absent: unpinned vs skimage).
    x = (o + 0.5) * in/out - 0.5; i = floor(x); out = (1 - f) v[m(i)] + f v[m(i+1)], m = mirror about the edge samples."""
    img = np.asarray(img, np.float64)
    h, w = img.shape
    oh, ow = int(round(h * scale)), int(round(w * scale))

    def axis(n, on):
        x = (np.arange(on) + 0.5) * (n / on) - 0.5
        i = np.floor(x).astype(np.int64)
        f = x - i
        period = max(2 * n - 2, 1)

        def mirror(k):
            k = np.mod(k, period)
            return np.where(k < n, k, period - k) if n > 1 else np.zeros_like(k)

        return mirror(i), mirror(i + 1), f

    y0, y1, fy = axis(h, oh)
    x0, x1, fx = axis(w, ow)
    rows = (1 - fy)[:, None] * img[y0, :] + fy[:, None] * img[y1, :]
    return (1 - fx)[None, :] * rows[:, x0] + fx[None, :] * rows[:, x1]


def calculate_contrast(cancer_loc, contralateral_loc, noise_loc, scale, image, focus):
    """nn_mri.py:59-85 restated (C, CNR, CNR2 of 2*scale-wide squares); pinned by tests/golden/contrast.npz."""
    def area(loc):
        x, y = ((int(i) - focus) * scale for i in loc)
        return np.asarray(image)[x - scale:x + scale, y - scale:y + scale]
    ca, cb, cn = area(cancer_loc), area(contralateral_loc), area(noise_loc)
    diff = abs(ca.mean() - cb.mean())
    return ca.mean() / (cb.mean() + 1e-7), diff / np.sqrt(np.std(ca) ** 2 + np.std(cb) ** 2), diff / np.std(cn)
