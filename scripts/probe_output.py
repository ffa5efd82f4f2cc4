"""Dev probe: relighting one 800x800 view under 16 probes + the asynchronous output path (Model.vis_batch)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict
from vqnerf_release_amd.decomp.nerfactor.util import vis
dev = torch.device('cuda:0')
rng = np.random.default_rng(1)
model = get_model_class('vq_nfr')(config_from_dict(bench.DECOMP_INI))
model.build_nets(device=dev, seed=0).to(dev)
cb = rng.uniform(0, 1, (15, 256)).astype(np.float32)
model.set_codebook(cb / np.linalg.norm(cb, axis=1, keepdims=True))
model.set_light(rng.uniform(0, 1, (16, 32, 3)).astype(np.float32))
model.novel_probes = {f'probe{i:02d}': torch.tensor(rng.uniform(0, 2, (16, 32, 3)).astype(np.float32)).to(dev) for i in range(16)}
H = W = 800
n = H * W
xyz = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
nrm = torch.nn.functional.normalize(xyz + 0.1 * torch.randn(n, 3, device=dev), dim=-1)
one = torch.ones(n, 1, device=dev)
batch = (['v'], torch.tensor([[H, W]], device=dev).repeat(n, 1), torch.tensor([[0, 0, 4.0]], device=dev).repeat(n, 1), torch.zeros(n, 3, device=dev),
         torch.rand(n, 3, device=dev), one, one.clone(), xyz, nrm, (torch.rand(n, 512, device=dev) < 0.7).float())
out = tempfile.mkdtemp()
w = vis.AsyncWriter(n_threads=int(sys.argv[1]) if len(sys.argv) > 1 else 12)
with torch.no_grad():
    for _ in range(2):
        model.fast_render(batch, mode='test', relight_probes=True)
    torch.cuda.synchronize()
    V = 6
    t0 = time.perf_counter()
    for v in range(V):
        _, _, _, to_vis = model.fast_render(batch, mode='test', relight_probes=True)
    torch.cuda.synchronize()
    t_render = (time.perf_counter() - t0) / V
    t0 = time.perf_counter()
    for v in range(V):
        _, _, _, to_vis = model.fast_render(batch, mode='test', relight_probes=True)
        model.vis_batch(to_vis, os.path.join(out, 'a%d' % v), mode='test', writer=w)
    torch.cuda.synchronize()
    t_loop = (time.perf_counter() - t0) / V
    w.flush()
    t_total = (time.perf_counter() - t0) / V
    # synchronous: render, then write this view to the end before the next one
    t0 = time.perf_counter()
    for v in range(2):
        _, _, _, to_vis = model.fast_render(batch, mode='test', relight_probes=True)
        model.vis_batch(to_vis, os.path.join(out, 's%d' % v), mode='test', writer=w).flush()
    t_sync = (time.perf_counter() - t0) / 2
files = os.listdir(os.path.join(out, 'a0'))
print(f'render only {t_render*1e3:.1f} ms/view; render + queueing the output {t_loop*1e3:.1f} ms/view (GPU loop); '
      f'until all files are on disk {t_total*1e3:.1f} ms/view; write-then-continue {t_sync*1e3:.1f} ms/view; {len(files)} files/view')
