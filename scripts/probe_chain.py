"""Dev probe: per-program timing of the reflectance MLP stacks (encoder, heads main, heads vq) on the f32 and split-precision
chain kernels.  VQN_LIB selects a diagnostic build of the library (see csrc/Makefile `diag`)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict

dev = torch.device('cuda:0')
model = get_model_class('vq_nfr')(config_from_dict(bench.DECOMP_INI))
model.build_nets(device=dev, seed=0).to(dev)
N = int(os.environ.get('PROBE_N', 640000))
xyz = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
z = torch.rand(N, 256, device=dev)
macs = {'enc': 179968, 'main': 296832, 'vq': 297600}


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for mode in sys.argv[1:] or ['f32', 'f16s']:
        model.matrix_mode = mode
        r = {'enc': t(lambda: model._pred_enc_at(xyz)), 'main': t(lambda: model._all_heads(z, 'main')), 'vq': t(lambda: model._all_heads(z, 'vq'))}
        print(os.environ.get('VQN_LIB', 'default').split('/')[-1], mode,
              ' '.join(f'{k} {v:.3f} ms ({2 * macs[k] * N / v / 1e9:.0f} TF)' for k, v in r.items()), f'total {sum(r.values()):.3f} ms', flush=True)
