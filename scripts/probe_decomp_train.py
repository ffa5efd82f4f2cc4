"""Dev probe: per-kernel time of the decomp training step at a large batch (rocprofv3 --kernel-trace --stats around it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vqnerf_release_amd.decomp.nerfactor import train_nfr
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict
dev = torch.device('cuda:0')
rng = np.random.default_rng(1)
stage3 = len(sys.argv) > 3 and sys.argv[3] == 'ref_nfr'     # the stage-3 model: frozen stage-2 parts, rgb_enc + 512-wide heads trained
ini = dict(bench.DECOMP_INI, model='ref_nfr') if stage3 else bench.DECOMP_INI
model = get_model_class(ini['model'])(config_from_dict(ini))
model.build_nets(device=dev, seed=0).to(dev)
if stage3:
    for name in ('fine_enc', 'bottleneck', 'spec_out'):
        for prm in model.net[name].parameters():
            prm.requires_grad_(False)
    model.register_trainable()
else:
    cb = rng.uniform(0, 1, (15, 256)).astype(np.float32)
    model.set_codebook(cb / np.linalg.norm(cb, axis=1, keepdims=True))
model.set_light(rng.uniform(0, 1, (16, 32, 3)).astype(np.float32))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
xyz = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
nrm = torch.nn.functional.normalize(xyz + 0.1 * torch.randn(n, 3, device=dev), dim=-1)
one = torch.ones(n, 1, device=dev)
batch = (['v'], torch.zeros(n, 2, device=dev), torch.tensor([[0, 0, 4.0]], device=dev).repeat(n, 1), torch.zeros(n, 3, device=dev),
         torch.rand(n, 3, device=dev), one, one.clone(), xyz, nrm, (torch.rand(n, 512, device=dev) < 0.7).float())
if stage3:
    batch = batch[:9] + (torch.rand(n, 3, device=dev),) + batch[9:]
else:
    model.get_codebook()
_ = model.light
graph = len(sys.argv) > 3 and sys.argv[3] == 'graph'        # the captured step replayed (kernel trace of what one replay launches)
if graph:
    opt, _, clip = train_nfr.make_optimizer(config_from_dict(ini), model.trainable_variables, capturable=True)
    tr = train_nfr.Trainer(model, opt, clip=clip, graph=True)
else:
    opt, _, clip = train_nfr.make_optimizer(config_from_dict(ini), model.trainable_variables)
    tr = train_nfr.Trainer(model, opt, clip=clip)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    tr.train_iter(batch, global_bs=n)
torch.cuda.synchronize()
print('done')
