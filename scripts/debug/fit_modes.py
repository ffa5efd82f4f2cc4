"""train_nfr.fit four ways on the two-view test set: graph=True twice, default, eager -- with and without code dropout: which histories agree?"""
import sys, os, tempfile
sys.path.insert(0, '.')
import numpy as np, torch
from pathlib import Path
from tests.test_datasets import _write_decomp_view, _decomp_cfg
from vqnerf_release_amd.decomp.nerfactor import train_nfr
from vqnerf_release_amd.decomp.nerfactor.datasets import get_dataset_class
tmp = Path(tempfile.mkdtemp())
rng = np.random.default_rng(3)
for vid in ('train_000', 'train_001', 'val_000'):
    _write_decomp_view(str(tmp / 'data'), str(tmp / 'geo'), vid, 24, 32, 512, rng, collapse=False)
for thres in ('0.2;0.4', '-'):
    cfg = _decomp_cfg(tmp, imh=24, n_rays_per_step=64, num_embed=6, num_drop=2, thres_str=thres, epochs=4, ckpt_period=4,
                      vali_period=0, total_sample_vq=64, random_seed=5, cluster_center_path='')
    tr = get_dataset_class('shape_unit')(cfg, 'train', device='cuda')
    res = {}
    for name, kw in (('graph_a', dict(graph=True)), ('graph_b', dict(graph=True)), ('default', {}), ('eager_a', dict(graph=False)), ('eager_b', dict(graph=False))):
        m, h = train_nfr.fit(cfg, str(tmp / f'run_{name}_{len(thres)}'), tr, None, log=lambda *_: None, **kw)
        res[name] = h['loss']
        print(thres, name, ['%.9f' % v for v in h['loss']], flush=True)
