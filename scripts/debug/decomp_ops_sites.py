"""Which Python lines issue the torch ops of one reflectance training step (2048 points)?  TorchDispatchMode + traceback."""
import sys, collections, traceback
sys.path.insert(0, '.')
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda:0')
model, tr, step = bench.decomp_train_setup(dev, 0, 1, graph=False)
for _ in range(4): step()
torch.cuda.synchronize()
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
with Log():
    step()
torch.cuda.synchronize()
by_site = collections.Counter()
for (s, n), c in sites.items():
    by_site[s] += c
print('torch ops in one step:', sum(sites.values()))
for s, c in by_site.most_common(40):
    ops = ', '.join(f'{n.replace("aten.", "")}x{k}' for (ss, n), k in sorted(sites.items(), key=lambda kv: -kv[1]) if ss == s)[:150]
    print(f'{c:4d}  {s:70s} {ops}')
