"""Dev probe: in-kernel phase breakdown of the training tile programs from the DIAGNOSTIC build (make -C vqnerf_release_amd/csrc stamps).
Usage:  VQN_LIB=vqnerf_release_amd/lib/libvqnerf_hip_stamps.so python scripts/probe_stamps_vm.py [rays]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vqnerf_release_amd import _C
from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork
from vqnerf_release_amd.geo.models.renderer import NeuSRenderer
from vqnerf_release_amd.geo.nerf_runner import SyntheticDataset
from vqnerf_release_amd.geo import train_programs as tp

assert 'stamps' in _C.LIB_PATH, 'set VQN_LIB to the stamps build'
dev = torch.device('cuda')
torch.manual_seed(0)
sdf, col, var = SDFNetwork(**bench.FULL['sdf']).to(dev), RenderingNetwork(**bench.FULL['color']).to(dev), SingleVarianceNetwork(0.3).to(dev)
ren = NeuSRenderer(None, sdf, var, col, **bench.FULL['renderer'])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
ds = SyntheticDataset(device=dev, n_images=8)
bg = torch.ones(1, 3, device=dev)
lib = _C.lib()
names = ['op decode + non-GEMM ops', 'GEMM tile set-up (aux fetches issued, acc init)', 'K loops (operand waits included)',
         'epilogues (act\', stash stores, LDS write)', 'barrier waits', 'total', 'workgroups']
orig = tp.NeusTrainEngine.run if hasattr(tp, 'NeusTrainEngine') else None
cls = [c for c in vars(tp).values() if isinstance(c, type) and hasattr(c, 'run') and hasattr(c, 'alloc_tensors')][0]
orig = cls.run


def run(self, which, *a, **k):
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.vqn_debug_read_stamps_vm(buf, 1)
    orig(self, which, *a, **k)
    torch.cuda.synchronize()
    lib.vqn_debug_read_stamps_vm(buf, 1)
    v = [int(x) for x in buf]
    print(which, f'wave-0 cycles per workgroup: {v[5] / max(v[6], 1):.3e} over {v[6]} workgroups')
    for i in range(5):
        print(f'    {names[i]:60s} {100.0 * v[i] / max(v[5], 1):5.1f} %')


def step():
    data = ds.gen_random_rays_at(0, B)
    o, d, rgb, mask = data[:, :3].contiguous(), data[:, 3:6].contiguous(), data[:, 6:9], data[:, 9:10]
    near, far = ds.near_far_from_sphere(o, d)
    r = ren.render(o, d, near, far, 2.0, background_rgb=bg, cos_anneal_ratio=1.0)
    loss = ((r['color_fine'] - rgb) * mask).abs().sum() / (mask.sum() + 1e-5) + 0.1 * r['gradient_error']
    loss.backward()


step()
cls.run = run
step()
