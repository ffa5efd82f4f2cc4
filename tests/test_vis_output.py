"""Output path (Model.vis_batch, vq_nfr.py:988-1134): files, compositing and metadata of one view, through the
asynchronous writer.  CPU: a stub model and host tensors; GPU: the real model after fast_render(relight_probes=True)."""
import json
import os

import numpy as np
import pytest
import torch

PIL = pytest.importorskip('PIL.Image')


class _Stub:
    white_bg = True
    light_res = (16, 32)
    novel_olat = {}

    def __init__(self):
        self.novel_probes = {'city': None, 'forest': None}
        self.light = torch.rand(16, 32, 3)

    def _validate_mode(self, mode):
        if mode not in ('train', 'vali', 'test', 'render'):
            raise ValueError(mode)

    def get_codebook(self):
        return torch.rand(256, 15)


def _view(H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    N = H * W
    R = lambda *s: torch.rand(*s, generator=g)
    alpha = (R(N, 1) > 0.3).float() * (0.5 + 0.5 * R(N, 1))
    d = {'id': ['val_007'], 'hw': torch.tensor([[H, W]] * N), 'gt_alpha': alpha, 'pred_alpha': alpha.clone(), 'gt_rgb': R(N, 3),
         'pred_rgb': R(N, 3) * 1.2 - 0.1, 'pred_vq_rgb': R(N, 3), 'pred_albedo': R(N, 3), 'pred_rough': R(N, 1), 'pred_normal': R(N, 3) * 2 - 1,
         'gt_normal': R(N, 3) * 2 - 1, 'pred_embed': torch.randint(0, 16, (N, 1)).float(), 'pred_rgb_probes': R(N, 2, 3), 'enc_z': R(N, 8),
         'pred_xyz': R(N, 3), 'pred_nothing': None}
    return d


def test_vis_batch_files_and_compositing(tmp_path):
    from vqnerf_release_amd.decomp.nerfactor.util import vis
    H, W = 6, 9
    m, d = _Stub(), _view(H, W)
    out = tmp_path / 'vis' / 'batch000'
    full = tmp_path / 'full'
    w = vis.vis_batch(m, d, str(out), mode='vali', alpha_thres=0.8, full_vis_path=str(full))
    assert vis.vis_batch(m, d, str(tmp_path / 'never'), mode='train') is not None and not (tmp_path / 'never').exists()
    with pytest.raises(ValueError):
        vis.vis_batch(m, d, str(out), mode='bogus')
    w.flush()
    names = set(os.listdir(out))
    want = {'gt_rgb.png', 'pred_rgb.png', 'pred_vq_rgb.png', 'pred_albedo.png', 'pred_albedo.npy', 'pred_rough.png', 'pred_rough.npy',
            'pred_normal.png', 'gt_normal.png', 'embed_map.png', 'pred_rgb_probes_city.png', 'pred_rgb_probes_forest.png',
            'gt_alpha.png', 'pred_alpha.png', 'metadata.json'}
    assert names == want, names ^ want
    assert set(os.listdir(full)) == {'vq_embed.npy', 'pred_embed.npy', 'enc_z.npy', 'pred_xyz.npy'}
    assert os.path.exists(tmp_path / 'vis' / 'pred_light.png') and np.load(tmp_path / 'vis' / 'np_light.npy').shape == (16, 32, 3)
    assert np.asarray(PIL.open(tmp_path / 'vis' / 'pred_light.png')).shape == (256, 512, 3)
    # compositing: thresholded gt alpha over white, clip, truncating 8-bit cast
    a = d['gt_alpha'].numpy().reshape(H, W).copy(); a[a < 0.8] = 0
    rgb = d['pred_rgb'].numpy().reshape(H, W, 3)
    img = rgb * a[..., None] + 1.0 * (1 - a[..., None])
    np.testing.assert_array_equal(np.asarray(PIL.open(out / 'pred_rgb.png')), (np.clip(img, 0, 1) * 255).astype(np.uint8))
    nrm = (d['pred_normal'].numpy().reshape(H, W, 3) + 1) / 2
    np.testing.assert_array_equal(np.asarray(PIL.open(out / 'pred_normal.png')),
                                  (np.clip(nrm * a[..., None] + (1 - a[..., None]), 0, 1) * 255).astype(np.uint8))
    np.testing.assert_array_equal(np.asarray(PIL.open(out / 'pred_albedo.png')),
                                  (d['pred_albedo'].numpy().reshape(H, W, 3) * 255).astype(np.uint8))      # as it is
    np.testing.assert_array_equal(np.load(out / 'pred_rough.npy'), d['pred_rough'].numpy().reshape(H, W))
    prb = d['pred_rgb_probes'].numpy().reshape(H, W, 2, 3)[:, :, 1]
    np.testing.assert_array_equal(np.asarray(PIL.open(out / 'pred_rgb_probes_forest.png')),
                                  (np.clip(prb * a[..., None] + (1 - a[..., None]), 0, 1) * 255).astype(np.uint8))
    emb = np.asarray(PIL.open(out / 'embed_map.png'))
    e = d['pred_embed'].numpy().reshape(H, W).astype(int)
    assert (emb[e == 0] == 0).all() and (emb[e == 1] == [0, 0, 255]).all() and (emb[e == 3] == [255, 0, 0]).all()
    meta = json.load(open(out / 'metadata.json'))
    g8, p8 = np.asarray(PIL.open(out / 'gt_rgb.png')), np.asarray(PIL.open(out / 'pred_rgb.png'))
    assert meta['id'] == 'val_007' and abs(meta['psnr'] - vis.psnr_uint8(g8, p8)) < 1e-9
    # test mode: no ground truth -> id only; simp: no metadata, no raw images
    w = vis.vis_batch(m, d, str(tmp_path / 't'), mode='test'); w.flush()
    assert json.load(open(tmp_path / 't' / 'metadata.json')) == {'id': 'val_007'}
    assert 'pred_xyz.npy' in os.listdir(tmp_path / 't') and 'enc_z.npy' not in os.listdir(tmp_path / 't')
    w = vis.vis_batch(m, d, str(tmp_path / 's'), mode='vali', simp=True); w.flush()
    assert 'metadata.json' not in os.listdir(tmp_path / 's') and 'gt_alpha.png' not in os.listdir(tmp_path / 's')


def test_async_writer_many_views_and_errors(tmp_path):
    from vqnerf_release_amd.decomp.nerfactor.util import vis
    m = _Stub()
    w = vis.AsyncWriter(n_threads=3)
    for i in range(12):                                          # more views than coordinator + worker threads
        vis.vis_batch(m, _view(5, 4, seed=i), str(tmp_path / ('v%02d' % i)), mode='test', writer=w)
    w.flush()
    assert all(os.path.exists(tmp_path / ('v%02d' % i) / 'pred_rgb.png') for i in range(12))
    bad = _view(5, 4)
    bad['pred_weird_key_q'] = torch.zeros(20, 3)
    vis.vis_batch(m, bad, str(tmp_path / 'bad'), mode='test', writer=w)
    with pytest.raises(NotImplementedError):
        w.flush()


@pytest.mark.gpu
def test_vis_batch_after_relighting_on_device(tmp_path):
    from oracle import decomp as od
    from tests.decomp_util import make_config, load_oracle_params, make_batch
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, _ = od.make_model_params(seed=0, K=15)
    model = load_oracle_params(get_model_class('vq_nfr')(make_config()), p, 'cuda')
    rng = np.random.default_rng(2)
    model.novel_probes = {f'probe{i:02d}': torch.tensor(rng.uniform(0, 2, (16, 32, 3)).astype(np.float32)).cuda() for i in range(3)}
    H, W = 16, 24
    batch = list(make_batch(od.make_points(H * W, seed=5), 'cuda'))
    batch[1] = torch.tensor([[H, W]] * (H * W)).cuda()
    with torch.no_grad():
        pred, gt, _, to_vis = model.fast_render(tuple(batch), mode='test', relight_probes=True)
    w = model.vis_batch(to_vis, str(tmp_path / 'view'), mode='test')
    w.flush()
    files = set(os.listdir(tmp_path / 'view'))
    assert {'pred_rgb_probes_probe00.png', 'pred_rgb_probes_probe02.png', 'metadata.json'} <= files
    a = to_vis['gt_alpha'].cpu().numpy().reshape(H, W).copy(); a[a < 0.8] = 0
    v = to_vis['pred_rgb_probes'].cpu().numpy().reshape(H, W, 3, 3)[:, :, 2]
    np.testing.assert_array_equal(np.asarray(PIL.open(tmp_path / 'view' / 'pred_rgb_probes_probe02.png')),
                                  (np.clip(v * a[..., None] + (1 - a[..., None]), 0, 1) * 255).astype(np.uint8))
