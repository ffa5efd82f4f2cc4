"""Dense stack with skip-concats: mirror of decomp/nerfvq_nfr3/nerfactor/networks/mlp.py:24-50 (+ seq.py:24-38).

Parameters keep the Keras layout (`kernel [in, out]`, `bias [out]`, y = x @ kernel + bias) and Keras' default
initialisers (glorot-uniform kernel, zero bias); like Keras, a layer is built on first use, or explicitly with
`build(in_dim)`.  `__call__` is the torch (autograd) statement; inference goes through the fused layer programs of
decomp/packing.py (see models/vq_nfr.py)."""
import math

import torch
import torch.nn as nn

_ACT = {None: None, 'relu': torch.relu, 'sigmoid': torch.sigmoid}


class Dense(nn.Module):
    def __init__(self, units, activation=None):
        super().__init__()
        if activation not in _ACT:
            raise NotImplementedError(activation)
        self.units, self.activation = units, activation
        self.trainable = True
        self.kernel = None
        self.bias = None

    def build(self, in_dim, device=None, generator=None):
        lim = math.sqrt(6.0 / (in_dim + self.units))
        k = (torch.rand(in_dim, self.units, generator=generator) * 2.0 - 1.0) * lim
        self.kernel = nn.Parameter(k.to(device) if device is not None else k)
        self.bias = nn.Parameter(torch.zeros(self.units, device=device))

    def forward(self, x):
        if self.kernel is None:
            self.build(x.shape[-1], device=x.device)
        y = x @ self.kernel + self.bias
        f = _ACT[self.activation]
        return y if f is None else f(y)


class Network(nn.Module):
    def __init__(self, widths, act=None, skip_at=None):
        super().__init__()
        depth = len(widths)
        if act is None:
            act = [None] * depth
        assert len(act) == depth, 'If not `None`, `act` must have the same length as `widths`'
        self.widths, self.act = list(widths), list(act)
        self.layers = nn.ModuleList([Dense(w, a) for w, a in zip(widths, act)])
        self.skip_at = skip_at

    def in_dims(self, d_in):
        dims, d = [], d_in
        for i, w in enumerate(self.widths):
            dims.append(d)
            d = w + (d_in if self.skip_at is not None and i in self.skip_at else 0)
        return dims

    def build(self, d_in, device=None, generator=None):
        for layer, d in zip(self.layers, self.in_dims(d_in)):
            if layer.kernel is None:
                layer.build(d, device=device, generator=generator)
        self.d_in = d_in
        return self

    def forward(self, x):
        y = x
        for i, layer in enumerate(self.layers):
            y = layer(y)
            if self.skip_at is not None and i in self.skip_at:
                y = torch.cat((y, x), -1)
        return y
