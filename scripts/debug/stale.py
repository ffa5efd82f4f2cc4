import torch, numpy as np, sys
sys.path.insert(0, '.')
from tests.decomp_util import make_config, load_oracle_params, make_batch
from oracle import decomp as od
from vqnerf_release_amd.decomp.nerfactor import train_nfr
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
p, specs = od.make_model_params(seed=0, K=15)
cfg = make_config(n_rays_per_step=128, lr=5e-3)
batches = [make_batch(od.make_points(256, seed=40 + i), 'cuda') for i in range(10)]
view = make_batch(od.make_points(300, seed=99), 'cuda', bg_every=5)
runs = {}
for graph in (False, True):
    model = load_oracle_params(get_model_class('vq_nfr')(cfg), p, 'cuda')
    model.get_codebook(); _ = model.light
    opt, _, clip = train_nfr.make_optimizer(cfg, model.trainable_variables, capturable=True)
    tr = train_nfr.Trainer(model, opt, clip=clip, graph=graph)
    valis = []
    for i, b in enumerate(batches):
        v0 = model.net['fine_enc'].layers[0].kernel._version
        tr.train_iter(b, global_bs=256)
        print(graph, i, 'version', v0, '->', model.net['fine_enc'].layers[0].kernel._version, 'cb', model._codebook._version)
        if i in (4, 9):
            model.assume_foreground = False
            with torch.no_grad():
                pred, gt, lk, _ = model.call(view, mode='vali')
            d = {k: pred[k].clone() for k in ('rgb', 'albedo', 'rough', 'vq_rgb', 'vq_albedo', 'embed')}
            d['light'] = model.light.detach().clone()
            d['cb'] = model._codebook.detach().clone()
            for n, q in model.named_parameters():
                d['P:' + n] = q.detach().clone()
            valis.append(d)
            model.assume_foreground = graph
    runs[graph] = valis
for vi, (a, b) in enumerate(zip(runs[False], runs[True])):
    for k in a:
        if not torch.equal(a[k], b[k]):
            print('vali', vi, 'DIFF', k, float((a[k].float() - b[k].float()).abs().max()))
print('done')
