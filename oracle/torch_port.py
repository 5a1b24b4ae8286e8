"""PyTorch-CPU port of the reference INR fit loop -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Same un-fused op sequence the reference executes on its CPU path (``nn.Linear`` -> ``* omega``
-> ``sin`` per layer, autograd backward, ``torch.optim.Adam`` defaults, full-batch MSE), written
fresh so it can travel to the GPU box where ``/root/reference`` does not exist.  It is what
``bench.py`` times as ``cpu_baseline`` (kind "port") and what the ``-m gpu`` parity tests use as
the live checker for gradients and short trajectories.  Pinned against the real reference by
``tests/test_oracle_golden.py`` (fixtures from ``oracle/gen_golden.py``).  Never imported by the
product package.

Citations are relative to ``/root/reference/implicit-neural-representations``.
"""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn


def port_mgrid(shape):
    """SRDWI.py:12-18."""
    axes = [torch.linspace(-1, 1, steps=int(n)) for n in shape]
    return torch.stack(torch.meshgrid(*axes, indexing="ij"), dim=-1).reshape(-1, len(shape))


def port_input_mapping(x, B):
    """SRDWI.py:111-116."""
    if B is None:
        return x
    proj = torch.matmul(2.0 * np.pi * x, B.T)
    return torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)


class PortSine(nn.Module):
    """SRDWI.py:41-59: weight U(-1/in, 1/in) if first else U(+-sqrt(6/in)/omega); default bias."""

    def __init__(self, fan_in, fan_out, first, omega):
        super().__init__()
        self.omega_0 = omega
        self.linear = nn.Linear(fan_in, fan_out)
        bound = 1.0 / fan_in if first else math.sqrt(6.0 / fan_in) / omega
        with torch.no_grad():
            self.linear.weight.uniform_(-bound, bound)

    def forward(self, x):
        return torch.sin(self.omega_0 * self.linear(x))


class PortSiren(nn.Module):
    """``Siren`` with the reference's RNG draw order.

    ``flavor='SRDWI'``: head constructed FIRST (SRDWI.py:75), clones+detaches input (SRDWI.py:88).
    ``flavor='INRmodel'``: head constructed LAST (INRmodel.py:140), no detach (INRmodel.py:147).
    Parameter/state-dict names match the reference: ``final_linear.*``, ``net.k.linear.*``.
    """

    def __init__(self, in_features, hidden_features, hidden_layers, out_features,
                 first_omega_0=30.0, hidden_omega_0=30.0, flavor="SRDWI"):
        super().__init__()
        head_bound = math.sqrt(6.0 / hidden_features) / hidden_omega_0
        self.flavor = flavor

        def make_head():
            head = nn.Linear(hidden_features, out_features)
            with torch.no_grad():
                head.weight.uniform_(-head_bound, head_bound)
            return head

        if flavor == "SRDWI":
            self.final_linear = make_head()
        layers = [PortSine(in_features, hidden_features, True, first_omega_0)]
        for _ in range(hidden_layers):
            layers.append(PortSine(hidden_features, hidden_features, False, hidden_omega_0))
        if flavor != "SRDWI":
            self.final_linear = make_head()
        layers.append(self.final_linear)
        self.net = nn.Sequential(*layers)

    def forward(self, coords):
        if self.flavor == "SRDWI":
            coords = coords.clone().detach().requires_grad_(False)
        return self.net(coords)

    def layer_params(self):
        """Weights and biases in network order (sine layers then head) as numpy arrays."""
        ws, bs = [], []
        for mod in self.net:
            lin = mod.linear if isinstance(mod, PortSine) else mod
            ws.append(lin.weight.detach().numpy().copy())
            bs.append(lin.bias.detach().numpy().copy())
        return ws, bs


class PortPN(nn.Module):
    """SRDWI.py:93-109 without the hard-coded ``.cuda()`` (SRDWI.py:102)."""

    def __init__(self, in_features, hidden_features, dimension):
        super().__init__()
        self.perturb_linear = nn.Linear(in_features + 1, hidden_features)
        self.perturb_linear2 = nn.Linear(hidden_features, dimension)

    def forward(self, coords, sample=0, eps=0.0):
        coords = coords.clone().detach()
        acq = torch.tensor([sample / 10.0], dtype=torch.float).repeat(coords.size(0), 1)
        h = torch.tanh(self.perturb_linear(torch.cat((coords, acq), -1)))
        return eps * torch.tanh(self.perturb_linear2(h))


def port_fit(model, model_input, target, steps, lr=1e-4, weight=None, optimizer=None, on_step=None):
    """The reference loop superresDWI.py:132-138 (weighted form master.py:143-148)."""
    opt = optimizer or torch.optim.Adam(lr=lr, params=list(model.parameters()))
    losses = []
    for k in range(steps):
        out = model(model_input)
        sq = (out - target) ** 2
        loss = (sq if weight is None else weight * sq).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
        if on_step is not None:
            on_step(k + 1, model)
    return losses, opt


def port_reconstruct(model, shape, B, clamp=True, chunk=262144):
    """superresDWI.py:125-126,161: clamp(INR(input_mapping(get_mgrid(shape), B)), 0).view(shape)."""
    grid = port_mgrid(shape)
    outs = []
    with torch.no_grad():
        for s in range(0, grid.shape[0], chunk):
            outs.append(model(port_input_mapping(grid[s:s + chunk], B)))
    y = torch.cat(outs, 0)
    if clamp:
        y = torch.clamp(y, min=0)
    return y.view(*shape).numpy()


def synthetic_volume(side=128, seed=0):
    """``default_rng(seed).random((side,)*3)`` -> fp32."""
    return np.random.default_rng(seed).random((side, side, side)).astype(np.float32)


def fourier_matrix(dim, mapping_size=128, scale=0.5, seed=0):
    """superresDWI.py:102-106 with a fixed seed."""
    np.random.seed(seed)
    return (np.random.normal(size=(mapping_size, dim)) * scale).astype(np.float32)
