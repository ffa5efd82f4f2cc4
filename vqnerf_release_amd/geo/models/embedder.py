"""Positional encoding, host-side mirror of geo/NeuS-ours2/models/embedder.py:6-51 (same
`get_embedder(multires, input_dims)` -> (embed_fn, out_dim) contract).  The HIP kernels compute the
same features in-register (csrc/mlp_prims.h: posenc_feat); this torch form serves the autograd path."""
import torch


class Embedder:
    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs['input_dims']
        n = kwargs['num_freqs']
        if kwargs['log_sampling']:
            self.freq_bands = 2.0 ** torch.linspace(0.0, kwargs['max_freq_log2'], n)
        else:
            self.freq_bands = torch.linspace(2.0 ** 0.0, 2.0 ** kwargs['max_freq_log2'], n)
        self.out_dim = (d if kwargs['include_input'] else 0) + d * n * len(kwargs['periodic_fns'])

    def embed(self, inputs):
        parts = [inputs] if self.kwargs['include_input'] else []
        for f in self.freq_bands.tolist():
            for fn in self.kwargs['periodic_fns']:
                parts.append(fn(inputs * f))
        return torch.cat(parts, -1)


def get_embedder(multires, input_dims=3):
    eo = Embedder(include_input=True, input_dims=input_dims, max_freq_log2=multires - 1, num_freqs=multires,
                  log_sampling=True, periodic_fns=[torch.sin, torch.cos])
    return (lambda x, eo=eo: eo.embed(x)), eo.out_dim
