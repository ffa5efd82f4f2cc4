"""Model base: mirror of decomp/nerfvq_nfr3/nerfactor/models/base.py:25-143 (registry glue the trainers rely on:
`net` dict, `register_trainable()`, `trainable_variables`, `_validate_mode`, `call`/`compute_loss` contract)."""
import torch.nn as nn


class Model(nn.Module):
    def __init__(self, config, debug=False):
        super().__init__()
        self.config = config
        self.debug = debug
        self.net = {}
        self.trainable_registered = False

    def register_trainable(self):
        """Sub-networks live in the plain dict `self.net`; alias every trainable layer directly under `self`
        (`net_<name>_layer<i>`) so that it is tracked -- the reference does the same for Keras (base.py:81-104)."""
        registered = []
        for net_name, net in self.net.items():
            attr = 'net_' + net_name
            assert attr.isidentifier(), net_name
            for i, layer in enumerate(net.layers):
                if getattr(layer, 'trainable', True):
                    full = f'{attr}_layer{i}'
                    if not hasattr(self, full):
                        setattr(self, full, layer)
                    registered.append(full)
        self.trainable_registered = True
        return registered

    @property
    def trainable_variables(self):
        return [p for p in self.parameters() if p.requires_grad]

    @staticmethod
    def _validate_mode(mode):
        if mode not in ('train', 'vali', 'test', 'render'):
            raise ValueError(mode)

    def forward(self, *args, **kwargs):
        return self.call(*args, **kwargs)

    def call(self, batch, mode='train'):
        raise NotImplementedError

    def compute_loss(self, pred, gt, **kwargs):
        raise NotImplementedError
