"""Which Python lines issue copy-like torch ops in the graph-mode reflectance step (the step that gets captured)?"""
import sys, collections, traceback
sys.path.insert(0, '.')
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device('cuda:0')
model, tr, step = bench.decomp_train_setup(dev, 0, 1, graph=True)
class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.sites = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(s in name for s in ('copy_', 'clone', 'contiguous', '_to_copy', 'cat', 'fill_', 'zero_', 'zeros', 'ones', 'full')):
            site = []
            for fr in reversed(traceback.extract_stack()):
                if 'vqnerf_release_amd/' in fr.filename:
                    site.append(f"{fr.filename.split('vqnerf_release_amd/')[-1]}:{fr.lineno}")
                    if len(site) == 2: break
            self.sites[(' <- '.join(site) or 'autograd-engine', name)] += 1
        return func(*args, **(kwargs or {}))
for it in range(4):
    with Log() as lg:
        step()
    torch.cuda.synchronize()
    print('--- step', it, 'captured' , getattr(tr, '_cap', None) is not None, 'ops', sum(lg.sites.values()))
    if it <= 1:
        for (s, n), c in lg.sites.most_common(40):
            print(f'{c:4d}  {n:32s} {s}')
