"""Which Python lines issue the torch ops INSIDE the captured reflectance training step (2048 points)?  TorchDispatchMode + traceback
around the capturing call of Trainer(graph=True); also the per-kernel launch census of one replay (torch profiler)."""
import sys, collections, traceback
sys.path.insert(0, '.')
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda:0')
model, tr, step = bench.decomp_train_setup(dev, 0, 1, graph=True)
sites = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in ('view', 'reshape', 'detach', 'alias', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'as_strided', 't.default', 'transpose', 'permute', 'empty', 'size', 'stride', '_local_scalar')):
            site = 'autograd-engine'
            for fr in reversed(traceback.extract_stack()):
                if 'vqnerf_release_amd/' in fr.filename:
                    site = f"{fr.filename.split('vqnerf_release_amd/')[-1]}:{fr.lineno}"
                    break
            sites[(site, name)] += 1
        return func(*args, **(kwargs or {}))
# the backward bodies run on the autograd engine's thread, where the mode above is not active: run each of them under its own Log()
import vqnerf_release_amd.decomp.refl_train as _rt, vqnerf_release_amd.decomp.nerfactor.models.nfr_unit as _nu
import vqnerf_release_amd.decomp.nerfactor.models.vq_nfr as _vn, vqnerf_release_amd.decomp.nerfactor.networks.vq_layers as _vl
import vqnerf_release_amd.decomp.nerfactor.util.math as _um
def _wrap(cls):
    inner = cls.backward
    def bw(ctx, *g):
        with Log():
            return inner(ctx, *g)
    cls.backward = staticmethod(bw)
for mod in (_rt, _nu, _vn, _vl, _um):
    for name in dir(mod):
        c = getattr(mod, name)
        if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function and c.__module__ == mod.__name__:
            _wrap(c)
torch.autograd.set_multithreading_enabled(False)      # backward on THIS thread: the dispatch mode sees the engine's own ops too
n = 0
while tr._captured is None and n < 8:
    if n >= tr.GRAPH_WARMUP:
        with Log():
            step()
    else:
        step()
    n += 1
torch.cuda.synchronize()
print('captured after', n, 'calls; torch ops during the capturing call:', sum(sites.values()))
by_site = collections.Counter()
for (s, nm), c in sites.items():
    by_site[s] += c
for s, c in by_site.most_common(60):
    ops = ', '.join(f'{nm.replace("aten.", "")}x{k}' for (ss, nm), k in sorted(sites.items(), key=lambda kv: -kv[1]) if ss == s)[:170]
    print(f'{c:4d}  {s:60s} {ops}')
