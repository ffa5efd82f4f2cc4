"""Positional encoding: mirror of decomp/nerfvq_nfr3/nerfactor/networks/embedder.py:23-47 (same constructor
keywords, `out_dims`, call -> concat([x, sin(f0 x), cos(f0 x), sin(f1 x), ...])).  The fused kernels compute the same
features in registers (csrc/mlp_prims.h: posenc_feat); this torch form serves the autograd path."""
import torch


class Embedder:
    def __init__(self, incl_input=True, in_dims=3, log2_max_freq=3, n_freqs=4, log_sampling=True, periodic_func=None):
        if periodic_func is None:
            periodic_func = [torch.sin, torch.cos]
        if log_sampling:
            bands = 2.0 ** torch.linspace(0.0, float(log2_max_freq), n_freqs)
        else:
            bands = torch.linspace(2.0 ** 0.0, 2.0 ** float(log2_max_freq), n_freqs)
        self.freq_bands = [float(b) for b in bands]
        self.periodic_func = list(periodic_func)
        self.incl_input = incl_input
        self.in_dims = in_dims
        self.n_freqs = n_freqs
        self.log_sampling = log_sampling
        self.out_dims = (in_dims if incl_input else 0) + in_dims * n_freqs * len(self.periodic_func)

    def __call__(self, x):
        parts = [x] if self.incl_input else []
        for f in self.freq_bands:
            for fn in self.periodic_func:
                parts.append(fn(x * f))
        return torch.cat(parts, -1)

    def fused_ok(self):
        """True when the kernels' built-in posenc (powers of two, sin then cos, input included) is this embedder."""
        return (self.incl_input and self.in_dims == 3 and self.log_sampling
                and self.periodic_func == [torch.sin, torch.cos]
                and self.freq_bands == [float(2 ** k) for k in range(self.n_freqs)])
