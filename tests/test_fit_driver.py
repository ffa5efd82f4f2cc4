"""Epoch driver of the VQ stage (train_nfr.fit; reference main(), train_nfr.py:93-378): threshold schedule, main codebook
size selection, per-epoch metrics (CPU); the whole loop on a two-view set with checkpoints, resume and validation
output (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.decomp_util import make_config


def test_thres_schedule_and_main_vq_selection():
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    cfg = make_config(num_embed=8, num_drop=3, thres_str='0.1;0.2;0.4')
    tt, vl, xs = train_nfr.thres_schedule(cfg)
    np.testing.assert_allclose(tt, [0, 0, 0, 0, 0, 0.1, 0.2, 0.4])
    assert xs == [5, 6, 7, 8] and len(vl) == 4
    np.testing.assert_array_equal(vl[0], [0, 0, 0, 0, 0, 1, 1, 1])           # first: the fewest codes (5) ...
    np.testing.assert_array_equal(vl[-1], [0] * 8)                             # ... last: all 8
    tt2, _, _ = train_nfr.thres_schedule(make_config(num_embed=8, num_drop=3, thres_str='-'))
    assert len(tt2) == 5 and not tt2.any()
    sel = train_nfr.select_main_vq
    assert sel([0.5, 0.3, 0.29, 0.28], 0.05) == 1       # drops, and no later size is more than 0.05 better
    assert sel([0.5, 0.3, 0.2, 0.19], 0.05) == 2        # size 1 is beaten by > 0.05 later on; 2 is not
    assert sel([0.3, 0.5, 0.6, 0.7], 0.05) == 3         # never improves on its predecessor: the full codebook
    assert sel([0.5, 0.3, 0.2, 0.1], 0.05) == 3
    assert sel([0.4, 0.3], 0.05) == 1


def test_save_metas(tmp_path):
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    for e, vals in ((1, (30.0, 32.0)), (2, (35.0,))):
        for b, psnr in enumerate(vals):
            d = tmp_path / 'vis_vali' / ('epoch%09d' % e) / 'main_15' / ('batch%09d' % b)
            d.mkdir(parents=True)
            json.dump({'id': 'v', 'psnr': psnr}, open(d / 'metadata.json', 'w'))
    m = train_nfr.save_metas(str(tmp_path))
    assert m['psnr'] == [31.0, 35.0] and m['ssim'] == [None, None]
    assert json.load(open(tmp_path / 'vis_vali' / 'metas.json'))['psnr'] == [31.0, 35.0]


@pytest.mark.gpu
def test_fit_trains_checkpoints_resumes_and_validates(tmp_path):
    from tests.test_datasets import _write_decomp_view, _decomp_cfg
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.datasets import get_dataset_class
    rng = np.random.default_rng(3)
    for vid in ('train_000', 'train_001', 'val_000'):
        _write_decomp_view(str(tmp_path / 'data'), str(tmp_path / 'geo'), vid, 24, 32, 512, rng, collapse=False)
    cfg = _decomp_cfg(tmp_path, imh=24, n_rays_per_step=64, num_embed=6, num_drop=2, thres_str='0.2;0.4', epochs=4, ckpt_period=2,
                      vali_period=4, vali_batches=1, total_sample_vq=64, best_thres=0.01, keep_recent_epochs=1, random_seed=5,
                      cluster_center_path='')
    Dataset = get_dataset_class('shape_unit')
    tr, va = Dataset(cfg, 'train', device='cuda'), Dataset(cfg, 'vali', device='cuda')
    out = str(tmp_path / 'run')
    model, hist = train_nfr.fit(cfg, out, tr, va, epochs=2, log=lambda *_: None)
    assert len(hist['loss']) == 2 and all(np.isfinite(hist['loss'])) and hist['vali'] == []
    assert os.listdir(os.path.join(out, 'checkpoints')) == ['ckpt-2.pt'] and os.path.exists(os.path.join(out, 'cluster_init.npy'))
    cb2 = model.get_codebook().detach().clone()
    # resume: two more epochs from the checkpoint, then the validation pass with its output tree
    model2, hist2 = train_nfr.fit(cfg, out, tr, va, log=lambda *_: None)
    assert len(hist2['loss']) == 2 and os.listdir(os.path.join(out, 'checkpoints')) == ['ckpt-4.pt']
    assert not torch.equal(model2.get_codebook(), cb2)
    (v,) = hist2['vali']
    assert v['step'] == 4 and len(v['drop_losses']) == 3 and 0 <= v['main_vq'] <= 2 and len(v['vis_dirs']) == 3
    edir = os.path.join(out, 'vis_vali', 'epoch%09d' % 4)
    subs = sorted(d for d in os.listdir(edir) if os.path.isdir(os.path.join(edir, d)))
    assert len(subs) == 3 and sum(s.startswith('main_') for s in subs) == 1 and {s.replace('main_', '') for s in subs} == {'4', '5', '6'}
    for s in subs:
        files = os.listdir(os.path.join(edir, s, 'batch%09d' % 0))
        assert 'pred_vq_rgb.png' in files and 'pred_rgb.png' in files and 'embed_map.png' in files
    assert {'loss.json', 'vq_test_loss.json'} <= set(os.listdir(edir))
    assert os.path.exists(os.path.join(out, 'vis_vali', 'vis_params', 'epoch%09d' % 4, 'vq_embed.npy'))
    assert os.path.exists(os.path.join(edir, subs[0], 'pred_light.png')) and os.path.exists(os.path.join(out, 'vis_vali', 'metas.json'))
    # the same loop with the step replayed from a captured HIP graph (code-dropout thresholds as a graph input)
    cfg_g = _decomp_cfg(tmp_path, imh=24, n_rays_per_step=64, num_embed=6, num_drop=2, thres_str='0.2;0.4', epochs=4, ckpt_period=4,
                        vali_period=0, total_sample_vq=64, random_seed=5, cluster_center_path='')
    model_g, hist_g = train_nfr.fit(cfg_g, str(tmp_path / 'run_graph'), tr, None, graph=True, log=lambda *_: None)
    assert len(hist_g['loss']) == 4 and all(np.isfinite(hist_g['loss'])) and os.path.exists(tmp_path / 'run_graph' / 'checkpoints' / 'ckpt-4.pt')
    # the captured step is what a caller gets by default (graph=None: whenever eligible): the same run as graph=True ...
    assert train_nfr.graph_default('cuda', model_g) is True and train_nfr.graph_default('cpu') is False
    model_d, hist_d = train_nfr.fit(cfg_g, str(tmp_path / 'run_default'), tr, None, log=lambda *_: None)
    assert hist_d['loss'] == hist_g['loss']
    for (n1, a), (_, b) in zip(model_d.state_dict().items(), model_g.state_dict().items()):
        assert torch.equal(a, b), n1
    # ... and, without code dropout, the eager loop bit for bit (with dropout the two draw their thresholds' uniforms at different
    # offsets of the device generator: a captured step reserves its draws per replay)
    cfg_n = _decomp_cfg(tmp_path, imh=24, n_rays_per_step=64, num_embed=6, num_drop=2, thres_str='-', epochs=4, ckpt_period=4,
                        vali_period=0, total_sample_vq=64, random_seed=5, cluster_center_path='')
    model_n, hist_n = train_nfr.fit(cfg_n, str(tmp_path / 'run_default_nodrop'), tr, None, log=lambda *_: None)
    model_e, hist_e = train_nfr.fit(cfg_n, str(tmp_path / 'run_eager_nodrop'), tr, None, graph=False, log=lambda *_: None)
    assert hist_n['loss'] == hist_e['loss']
    for (n1, a), (_, b) in zip(model_n.state_dict().items(), model_e.state_dict().items()):
        assert torch.equal(a, b), n1
    # inference pass of test.py: every validation view relit under the probes of `test_envmap_dir` and the OLAT maps
    from vqnerf_release_amd.decomp.nerfactor.util import io as ioutil
    os.makedirs(tmp_path / 'probes')
    for name in ('city', 'forest'):
        ioutil.write_hdr(str(tmp_path / 'probes' / (name + '.hdr')), rng.uniform(0, 2, (16, 32, 3)).astype(np.float32))
    model2.config.set('DEFAULT', 'test_envmap_dir', str(tmp_path / 'probes'))
    model2._novel_lights()
    model2.to('cuda')
    # (reference behaviour: `relight_olat` is accepted and ignored, vq_nfr.py:733 -- no OLAT files)
    w, n = train_nfr.render_views(model2, va, str(tmp_path / 'pd_relit_ref'), relight_olat=True, relight_probes=True)
    w.flush()
    files = set(os.listdir(tmp_path / 'pd_relit_ref' / ('batch%09d' % 0)))
    assert n == 1 and {'pred_rgb_probes_city.png', 'pred_rgb_probes_forest.png', 'metadata.json'} <= files
    assert not any(f.startswith('pred_rgb_olat') for f in files)
    model2.render_olat = True                                 # the build's opt-in: OLAT maps in the same shading pass
    w, n = train_nfr.render_views(model2, va, str(tmp_path / 'pd_relit'), relight_olat=True, relight_probes=True)
    w.flush()
    files = set(os.listdir(tmp_path / 'pd_relit' / ('batch%09d' % 0)))
    assert n == 1 and {'pred_rgb_probes_city.png', 'pred_rgb_probes_forest.png', 'pred_rgb_olat_0004-0008.png', 'metadata.json'} <= files
    # view sharding of a multi-process inference run: process 1 of 2 gets views 1, 3, ... and keeps their global numbering
    w, n = train_nfr.render_views(model2, tr, str(tmp_path / 'sharded'), num_p=2, p_i=1)
    w.flush()
    assert n == 1 and os.listdir(tmp_path / 'sharded') == ['batch%09d' % 1]
    w, n = train_nfr.render_views(model2, tr, str(tmp_path / 'sharded'), num_p=2, p_i=0)
    w.flush()
    assert n == 1 and sorted(os.listdir(tmp_path / 'sharded')) == ['batch%09d' % 0, 'batch%09d' % 1]


def test_outer_sample_max_colour_difference_neighbour():
    """trainvali.py:327-336: every pixel is paired with the 8-neighbour whose colour differs most (first one on ties)."""
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    H, W = 9, 11
    rng = np.random.default_rng(0)
    rgb = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
    rgb[4, 5] = rgb[4, 4]                                            # a tie candidate
    N = H * W
    hw = torch.tensor([[H, W]] * N)
    flat_idx = torch.arange(N, dtype=torch.float32)[:, None]
    batch = (['v'], hw, flat_idx.clone(), torch.zeros(N, 3), torch.tensor(rgb.reshape(N, 3)), torch.ones(N, 1), torch.ones(N, 1),
             torch.zeros(N, 3), torch.zeros(N, 3), torch.zeros(N, 4))
    cfg = make_config(n_rays_per_step=200)
    out = train_nfr.outer_sample(batch, cfg, 'nerf', generator=torch.Generator().manual_seed(0), neighbour='max_diff')
    idx = out[2][:, 0].long().numpy()                                # rayo slot carries the flat pixel index
    jit = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]
    for p, q in zip(idx[0::2], idx[1::2]):
        y, x = divmod(int(p), W)
        d = [np.abs(rgb[y + dy, x + dx] - rgb[y, x]).max() for dy, dx in jit]
        k = int(np.argmax(d))
        assert 1 <= y < H - 1 and 1 <= x < W - 1 and int(q) == (y + jit[k][0]) * W + x + jit[k][1]


@pytest.mark.gpu
def test_fit_stage_trains_stage1_with_a_pretrain_epoch(tmp_path):
    from tests.test_datasets import _write_decomp_view, _decomp_cfg
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.datasets import get_dataset_class
    rng = np.random.default_rng(3)
    for vid in ('train_000', 'train_001', 'val_000'):
        _write_decomp_view(str(tmp_path / 'data'), str(tmp_path / 'geo'), vid, 24, 32, 512, rng, collapse=False)
    cfg = _decomp_cfg(tmp_path, imh=24, n_rays_per_step=64, model='nfr_unit', epochs=3, pretrain_epochs=2, ckpt_period=3, vali_period=3,
                      vali_batches=1, random_seed=2)
    Dataset = get_dataset_class('shape_unit')
    tr, va = Dataset(cfg, 'train', device='cuda'), Dataset(cfg, 'vali', device='cuda')
    model, hist = train_nfr.fit_stage(cfg, str(tmp_path / 'run1'), tr, va, log=lambda *_: None)
    # default = the captured step (two eager warm-up steps, a capture with pretrain=True, a second capture when the pretraining epochs
    # end): the same numbers as the eager loop, bit for bit -- validation views with background rows included
    model_e, hist_e = train_nfr.fit_stage(cfg, str(tmp_path / 'run1_eager'), tr, va, graph=False, log=lambda *_: None)
    assert hist['loss'] == hist_e['loss']
    for (n1, a), (_, b) in zip(model.state_dict().items(), model_e.state_dict().items()):
        assert torch.equal(a, b), n1
    with open(os.path.join(hist['vali_dirs'][0], 'metadata.json')) as f, open(os.path.join(hist_e['vali_dirs'][0], 'metadata.json')) as fe:
        assert json.load(f)['psnr'] == json.load(fe)['psnr']
    assert type(model).__module__.endswith('nfr_unit') and len(hist['loss']) == 3 and all(np.isfinite(hist['loss']))
    assert os.listdir(tmp_path / 'run1' / 'checkpoints') == ['ckpt-3.pt'] and len(hist['vali_dirs']) == 1
    files = set(os.listdir(hist['vali_dirs'][0]))
    assert {'pred_rgb.png', 'gt_rgb.png', 'pred_albedo.png', 'pred_albedo.npy', 'metadata.json'} <= files
    meta = json.load(open(os.path.join(hist['vali_dirs'][0], 'metadata.json')))
    assert meta['id'] == 'val_000' and np.isfinite(meta['psnr'])
    assert json.load(open(tmp_path / 'run1' / 'vis_vali' / 'metas.json'))['psnr'] == [meta['psnr']]


@pytest.mark.gpu
def test_fit_stage_trains_stage3_on_the_dedicated_kernels(tmp_path):
    """Stage 3 end to end: datasets/ref_nfr views from disk -> fit_stage (captured step by default) with the stage-2 parts frozen: the same
    numbers as the eager loop bit for bit, rgb_enc and the 512-wide heads move, the frozen parts do not, and no launch of the step is an
    interpreted tile program."""
    from tests.gpu_util import launches
    from tests.test_datasets import _write_decomp_view, _decomp_cfg
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.datasets import get_dataset_class
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    rng = np.random.default_rng(6)
    for vid in ('train_000', 'train_001', 'val_000'):
        _write_decomp_view(str(tmp_path / 'data'), str(tmp_path / 'geo'), vid, 24, 32, 512, rng, collapse=False)
    cfg = _decomp_cfg(tmp_path, imh=24, n_rays_per_step=64, model='ref_nfr', epochs=3, ckpt_period=3, vali_period=3, vali_batches=1, random_seed=2)
    Dataset = get_dataset_class('ref_nfr')
    tr, va = Dataset(cfg, 'train', device='cuda'), Dataset(cfg, 'vali', device='cuda')

    def stage3_model():
        m = get_model_class('ref_nfr')(cfg)
        m.build_nets(device='cuda', seed=2).to('cuda')
        for name in ('fine_enc', 'bottleneck', 'spec_out'):                # load_stage2's freeze
            for p in m.net[name].parameters():
                p.requires_grad_(False)
        return m
    init = {k: v.detach().clone() for k, v in stage3_model().state_dict().items()}
    with launches() as rec:
        model, hist = train_nfr.fit_stage(cfg, str(tmp_path / 'run3'), tr, va, model=stage3_model(), log=lambda *_: None)
    assert rec.ran('vqn_refl_train_fwd_x3') and rec.ran('vqn_refl_train_bwd_x3') and not rec.ran('vqn_tile_program')
    model_e, hist_e = train_nfr.fit_stage(cfg, str(tmp_path / 'run3_eager'), tr, va, model=stage3_model(), graph=False, log=lambda *_: None)
    assert len(hist['loss']) == 3 and all(np.isfinite(hist['loss'])) and hist['loss'] == hist_e['loss']
    moved = 0
    for (n1, a), (_, b) in zip(model.state_dict().items(), model_e.state_dict().items()):
        assert torch.equal(a, b), n1
        if any(n1.startswith(f'net_{f}_') for f in ('fine_enc', 'bottleneck', 'spec_out')):
            assert torch.equal(a, init[n1]), n1
        elif n1.startswith('net_'):
            moved += int(not torch.equal(a, init[n1]))
    assert moved == 18                                                      # rgb_enc (6 tensors) + two heads (12): they all train
    assert len(hist['vali_dirs']) == 1 and 'pred_rgb.png' in os.listdir(hist['vali_dirs'][0])
