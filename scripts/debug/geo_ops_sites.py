"""Which Python lines issue the torch ops of one eager geo training step (2560 rays)?  TorchDispatchMode + traceback, autograd on this thread."""
import sys, collections, traceback
sys.path.insert(0, '.')
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda:0')
runner, step = bench.geo_train_setup(dev, 0, 2560, graph=False)
for _ in range(4): step()
torch.cuda.synchronize()
sites = collections.Counter()
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in ('view', 'reshape', 'detach', 'alias', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'as_strided', 't.default', 'transpose', 'permute', 'empty', 'size', 'stride', '_local_scalar', 'unbind', 'split', 'narrow')):
            site = 'autograd-engine'
            for fr in reversed(traceback.extract_stack()):
                if 'vqnerf_release_amd/' in fr.filename or fr.filename.endswith('bench.py'):
                    site = f"{fr.filename.split('repo/')[-1]}:{fr.lineno}"
                    break
            numel = 0
            for a in list(args) + list((kwargs or {}).values()):
                if torch.is_tensor(a): numel = max(numel, a.numel())
            sites[(site, name)] += 1
            big[(site, name)] = max(big.get((site, name), 0), numel)
        return func(*args, **(kwargs or {}))
big = {}
torch.autograd.set_multithreading_enabled(False)
with Log():
    step()
torch.cuda.synchronize()
print('torch ops in one eager geo step:', sum(sites.values()))
for (s, n), c in sorted(sites.items(), key=lambda kv: -big[kv[0]])[:70]:
    print(f'{c:3d}  numel<= {big[(s, n)]:9d}  {s[:70]:70s} {n}')
