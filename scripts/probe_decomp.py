"""Dev probe: per-launch kernel times of vq_nfr.call (inference) on an 800x800 point set."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vqnerf_release_amd import _C
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict

dev = torch.device('cuda')
rng = np.random.default_rng(1)
model = get_model_class('vq_nfr')(config_from_dict(bench.DECOMP_INI))
model.build_nets(device=dev, seed=0).to(dev)
cb = rng.uniform(0, 1, (15, 256)).astype(np.float32)
model.set_codebook(cb / np.linalg.norm(cb, axis=1, keepdims=True))
model.set_light(rng.uniform(0, 1, (16, 32, 3)).astype(np.float32))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 640000
xyz = rng.uniform(-1, 1, (N, 3)).astype(np.float32); xyz /= np.linalg.norm(xyz, axis=1, keepdims=True)
T = lambda a: torch.tensor(a, device=dev)
one = torch.ones(N, 1, device=dev)
batch = (['v'], torch.zeros(N, 2, device=dev), T(np.tile(np.array([[0, 0, 4.0]], np.float32), (N, 1))), torch.zeros(N, 3, device=dev),
         torch.rand(N, 3, device=dev), one, one.clone(), T(xyz * 0.8), T(xyz.copy()), (torch.rand(N, 512, device=dev) < 0.7).float())
with torch.no_grad():
    for _ in range(2):
        model.call(batch, mode=os.environ.get('VQN_MODE', 'test'))
    torch.cuda.synchronize()
    _C.KernelClock.reset(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        model.call(batch, mode=os.environ.get('VQN_MODE', 'test'))
    e1.record(); torch.cuda.synchronize()
print(f'N={N}: {e0.elapsed_time(e1)/3:.2f} ms per call')
for k, v in _C.KernelClock.pairs.items():
    print(' ', k, ['%.2f' % a.elapsed_time(b) for a, b in v])
