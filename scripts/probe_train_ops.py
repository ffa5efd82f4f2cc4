"""Dev probe: torch-op level profile of the geo training step of bench.py (which torch ops make up the glue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork
from vqnerf_release_amd.geo.models.renderer import NeuSRenderer
from vqnerf_release_amd.geo.nerf_runner import SyntheticDataset
dev = torch.device('cuda:0')
torch.manual_seed(0)
sdf, col, var = SDFNetwork(**bench.FULL['sdf']).to(dev), RenderingNetwork(**bench.FULL['color']).to(dev), SingleVarianceNetwork(0.3).to(dev)
ren = NeuSRenderer(None, sdf, var, col, **bench.FULL['renderer'])
B = 2560
ds = SyntheticDataset(device=dev, n_images=8)
params = list(sdf.parameters()) + list(var.parameters()) + list(col.parameters())
opt = torch.optim.Adam(params, lr=5e-4)
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
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cuda_time_total', row_limit=22, max_name_column_width=70))
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=12, max_name_column_width=70))
