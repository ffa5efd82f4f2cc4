import sys
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import decomp as od
from oracle import geo as og
from vqnerf_release_amd.geo import train_programs as tp
from vqnerf_release_amd.decomp.nerfactor import train_nfr
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from tests.decomp_util import make_config, load_oracle_params, make_batch
from tests.test_gpu_neus_render import _build
res = {}
for key, batched in (('a0', False), ('a1', False), ('b0', True), ('b1', True)):
    tp.BATCHED_WGRAD[0] = batched
    torch.manual_seed(0)
    p, specs = od.make_model_params(seed=0, K=15)
    model = load_oracle_params(get_model_class('vq_nfr')(make_config(n_rays_per_step=256)), p, 'cuda')
    batch = make_batch(od.make_points(300, seed=11), 'cuda')
    model.get_codebook(); _ = model.light
    names = [n for n, v in model.named_parameters()] if hasattr(model, 'named_parameters') else None
    opt = torch.optim.SGD(model.trainable_variables, lr=0.0)
    tr = train_nfr.Trainer(model, opt)
    tr.train_iter(batch, global_bs=300)
    g_dec = [(tuple(v.shape), v.grad.detach().clone()) for v in model.trainable_variables if v.grad is not None]
    cfg, sdf, col, var, ren = _build('full')
    o, d, near, far = [torch.tensor(a).cuda() for a in og.make_rays(24, 5)]
    rr = ren.render(o, d, near, far, 2.0, perturb_overwrite=0, background_rgb=torch.ones(1, 3).cuda(), cos_anneal_ratio=1.0)
    (rr['color_fine'].sum() + rr['gradient_error']).backward()
    g_geo = [(n, q.grad.detach().clone()) for mname, m in (('sdf', sdf), ('col', col)) for n, q in m.named_parameters()]
    res[key] = (g_dec, g_geo)
def cmp(x, y, label):
    bad = 0
    for part in (0, 1):
        for i, ((na, a), (nb, b)) in enumerate(zip(res[x][part], res[y][part])):
            if not torch.equal(a, b):
                bad += 1
                print(label, 'part', part, i, na, 'max diff', float((a - b).abs().max()), 'scale', float(a.abs().max()))
    print(label, 'differing tensors:', bad)
cmp('a0', 'a1', 'per-weight vs per-weight')
cmp('b0', 'b1', 'batched vs batched')
cmp('a0', 'b0', 'per-weight vs batched')
