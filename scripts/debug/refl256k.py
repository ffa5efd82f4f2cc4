"""262,144-point reflectance training step: wall time and the per-kernel clock under VQN_REFL_TRAIN = x3 | prog."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, bench
from vqnerf_release_amd import _C
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
for mode in (sys.argv[2:] or ['x3', 'prog', 'x3']):
    os.environ['VQN_REFL_TRAIN'] = mode
    for _ in range(2): tr.train_iter(batch, global_bs=n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): tr.train_iter(batch, global_bs=n)
    torch.cuda.synchronize()
    print(f'VQN_REFL_TRAIN={mode} n={n}: {(time.perf_counter()-t0)/5*1e3:.2f} ms/step', flush=True)
    _C.KernelClock.reset(True)
    tr.train_iter(batch, global_bs=n); torch.cuda.synchronize()
    summ = _C.KernelClock.summary(); _C.KernelClock.reset(False)
    print('   clocked total', sum(v[1] for v in summ.values()))
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])[:14]:
        print('      ', k, v)
