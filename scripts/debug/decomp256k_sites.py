"""Which Python lines issue the torch ops of one eager 262,144-point reflectance training step, and on how many elements?"""
import os, sys, collections, traceback
sys.path.insert(0, '.')
import numpy as np, torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
from vqnerf_release_amd.decomp.nerfactor import train_nfr
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict
dev = torch.device('cuda:0'); rng = np.random.default_rng(1)
model = get_model_class('vq_nfr')(config_from_dict(bench.DECOMP_INI)); model.build_nets(device=dev, seed=0).to(dev)
cb = rng.uniform(0, 1, (15, 256)).astype(np.float32); model.set_codebook(cb / np.linalg.norm(cb, axis=1, keepdims=True))
model.set_light(rng.uniform(0, 1, (16, 32, 3)).astype(np.float32))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
xyz = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
nrm = torch.nn.functional.normalize(xyz + 0.1 * torch.randn(n, 3, device=dev), dim=-1)
one = torch.ones(n, 1, device=dev)
batch = (['v'], torch.zeros(n, 2, device=dev), torch.tensor([[0, 0, 4.0]], device=dev).repeat(n, 1), torch.zeros(n, 3, device=dev),
         torch.rand(n, 3, device=dev), one, one.clone(), xyz, nrm, (torch.rand(n, 512, device=dev) < 0.7).float())
model.get_codebook(); _ = model.light
opt, _, clip = train_nfr.make_optimizer(config_from_dict(bench.DECOMP_INI), model.trainable_variables)
tr = train_nfr.Trainer(model, opt, clip=clip)
for _ in range(3): tr.train_iter(batch, global_bs=n)
torch.cuda.synchronize()
sites, big = collections.Counter(), {}
class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(s in name for s in ('view', 'reshape', 'detach', 'alias', 'expand', 'slice', 'select', 'unsqueeze', 'squeeze', 'as_strided', 't.default', 'transpose', 'permute', 'empty', 'size', 'stride', '_local_scalar', 'unbind', 'split', 'narrow')):
            site = 'autograd-engine'
            for fr in reversed(traceback.extract_stack()):
                if 'vqnerf_release_amd/' in fr.filename:
                    site = f"{fr.filename.split('vqnerf_release_amd/')[-1]}:{fr.lineno}"
                    break
            numel = 0
            for a in list(args) + list((kwargs or {}).values()):
                if torch.is_tensor(a): numel = max(numel, a.numel())
            if numel >= 1000000 and 'add' in name:
                try:
                    node = torch._C._current_autograd_node()
                except Exception:
                    node = None
                print('BIG ADD', [tuple(a.shape) for a in args if torch.is_tensor(a)], 'node:', None if node is None else node.name(), flush=True)
            sites[(site, name)] += 1
            big[(site, name)] = max(big.get((site, name), 0), numel)
        return func(*args, **(kwargs or {}))
torch.autograd.set_multithreading_enabled(False)
with Log():
    tr.train_iter(batch, global_bs=n)
torch.cuda.synchronize()
print('torch ops in one eager step:', sum(sites.values()))
for (s, nm), c in sorted(sites.items(), key=lambda kv: -big[kv[0]])[:40]:
    print(f'{c:3d}  numel<= {big[(s, nm)]:10d}  {s[:62]:62s} {nm}')
