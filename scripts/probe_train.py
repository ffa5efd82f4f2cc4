"""Dev probe: per-kernel breakdown of one geo training step (not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vqnerf_release_amd import _C
from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork
from vqnerf_release_amd.geo.models.renderer import NeuSRenderer
from vqnerf_release_amd.geo.nerf_runner import SyntheticDataset

dev = torch.device('cuda')
torch.manual_seed(0)
sdf, col, var = SDFNetwork(**bench.FULL['sdf']).to(dev), RenderingNetwork(**bench.FULL['color']).to(dev), SingleVarianceNetwork(0.3).to(dev)
ren = NeuSRenderer(None, sdf, var, col, **bench.FULL['renderer'])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2560
ds = SyntheticDataset(device=dev, n_images=8)
opt = torch.optim.Adam(list(sdf.parameters()) + list(var.parameters()) + list(col.parameters()), lr=5e-4)
bg = torch.ones(1, 3, device=dev)

def step():
    data = ds.gen_random_rays_at(0, B)
    o, d, rgb, mask = data[:, :3].contiguous(), data[:, 3:6].contiguous(), data[:, 6:9], data[:, 9:10]
    near, far = ds.near_far_from_sphere(o, d)
    opt.zero_grad(set_to_none=True)
    r = ren.render(o, d, near, far, 2.0, background_rgb=bg, cos_anneal_ratio=1.0)
    loss = ((r['color_fine'] - rgb) * mask).abs().sum() / (mask.sum() + 1e-5) + 0.1 * r['gradient_error'] \
        + 0.1 * torch.nn.functional.binary_cross_entropy(r['weight_sum'].clip(1e-3, 1 - 1e-3), mask)
    loss.backward()
    opt.step()

for _ in range(2):
    step()
torch.cuda.synchronize()
_C.KernelClock.reset(True)
t0 = time.perf_counter()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
clk = _C.KernelClock.summary()
print(f'B={B}: {dt*1e3:.2f} ms/step  {B/dt:.0f} rays/s')
tot = 0
for k, (c, ms) in sorted(clk.items(), key=lambda kv: -kv[1][1]):
    print(f'  {k:40s} {c//n:4d} launches/step  {ms/n:8.3f} ms/step')
    tot += ms / n
print(f'  sum of timed kernels {tot:.2f} ms; rest (torch ops, host) {dt*1e3 - tot:.2f} ms')
