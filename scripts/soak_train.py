"""One-off soak: a few hundred optimisation steps of both trainers on synthetic data; prints loss trajectories and checks that
nothing goes non-finite (not a benchmark, not a quality claim)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vqnerf_release_amd.geo.nerf_runner import Runner, SyntheticDataset

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
text = open(os.path.join(HERE, 'tests', 'golden', 'neus_like.conf')).read().replace('./exp/', '/tmp/soak_exp/')
text = text.replace('warm_up_end = 5000', 'warm_up_end = 50').replace('batch_size = 64', 'batch_size = 512')
torch.manual_seed(0)
r = Runner(conf_text=text, case='soak', dataset=SyntheticDataset(n_images=8, H=128, W=128))
r.update_learning_rate()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = time.time()
hist = []
for it in range(n):
    st = r.train_step(r.dataset.gen_random_rays_at(it % 8, r.batch_size))
    if it % 25 == 24:
        hist.append(float(st['loss']))
torch.cuda.synchronize()
print('geo  : %d steps in %.1f s; loss every 25 steps:' % (n, time.time() - t0), ' '.join('%.4f' % v for v in hist))
assert all(np.isfinite(hist)) and hist[-1] < hist[0]
for p in list(r.sdf_network.parameters()) + list(r.color_network.parameters()):
    assert torch.isfinite(p).all()

from vqnerf_release_amd.decomp.nerfactor import train_nfr
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict
dev = torch.device('cuda')
rng = np.random.default_rng(1)
cfg = config_from_dict(bench.DECOMP_INI)
model = get_model_class('vq_nfr')(cfg)
model.build_nets(device=dev, seed=0).to(dev)
cb = rng.uniform(0, 1, (15, 256)).astype(np.float32)
model.set_codebook(cb / np.linalg.norm(cb, axis=1, keepdims=True))
model.set_light(rng.uniform(0.2, 1, (16, 32, 3)).astype(np.float32))
N = 2048
xyz = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1)
nrm = torch.nn.functional.normalize(xyz + 0.1 * torch.randn(N, 3, device=dev), dim=-1)
one = torch.ones(N, 1, device=dev)
target = (0.3 + 0.4 * (xyz * 0.5 + 0.5)).clamp(0, 1)                      # a smooth colour field to fit
batch = (['v'], torch.zeros(N, 2, device=dev), torch.tensor([[0, 0, 4.0]], device=dev).repeat(N, 1), torch.zeros(N, 3, device=dev),
         target, one, one.clone(), xyz, nrm, (torch.rand(N, 512, device=dev) < 0.7).float())
model.get_codebook(); _ = model.light
opt, sched, clip = train_nfr.make_optimizer(cfg, model.trainable_variables, capturable=True)
tr = train_nfr.Trainer(model, opt, clip=clip, sched=sched, graph=True)
thres = torch.tensor([0.0] * 10 + [0.1, 0.2, 0.3, 0.4, 0.5], device=dev)
t0 = time.time()
hist = []
for it in range(2 * n):
    wl, _, _ = tr.train_iter(batch, global_bs=1024, thres=thres)
    if it % 50 == 49:
        hist.append(float(wl))
torch.cuda.synchronize()
print('decomp (captured step, code dropout): %d steps in %.1f s; loss every 50 steps:' % (2 * n, time.time() - t0), ' '.join('%.4f' % v for v in hist))
assert all(np.isfinite(hist)) and hist[-1] < hist[0]
assert torch.isfinite(model._codebook).all()
print('ok')
