"""Dev probe: per-kernel time of one full-view vq_nfr.call (mode from VQN_MODE, default 'test': a render) -- rocprofv3
--kernel-trace --stats around it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict
dev = torch.device('cuda:0')
rng = np.random.default_rng(1)
K = int(os.environ.get('VQN_K', '15'))                         # codebook size (BASELINE configs[2]: 64)
model = get_model_class('vq_nfr')(config_from_dict(dict(bench.DECOMP_INI, num_embed=K)))
model.build_nets(device=dev, seed=0).to(dev)
cb = rng.uniform(0, 1, (K, 256)).astype(np.float32)
model.set_codebook(cb / np.linalg.norm(cb, axis=1, keepdims=True))
model.set_light(rng.uniform(0, 1, (16, 32, 3)).astype(np.float32))
n = 640000
xyz = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
nrm = torch.nn.functional.normalize(xyz + 0.1 * torch.randn(n, 3, device=dev), dim=-1)
one = torch.ones(n, 1, device=dev)
one[::5] = 0.0
batch = (['v'], torch.zeros(n, 2, device=dev), torch.tensor([[0, 0, 4.0]], device=dev).repeat(n, 1), torch.zeros(n, 3, device=dev),
         torch.rand(n, 3, device=dev), one, one.clone(), xyz, nrm, (torch.rand(n, 512, device=dev) < 0.7).float())
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
        model.call(batch, mode=os.environ.get('VQN_MODE', 'test'))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    e0.record()
    for _ in range(5):
        model.call(batch, mode=os.environ.get('VQN_MODE', 'test'))
    e1.record()
torch.cuda.synchronize()
print('done: %.3f ms per call (640,000 rows, 512,000 foreground)' % (e0.elapsed_time(e1) / 5))
