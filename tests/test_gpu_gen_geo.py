"""GPU: geometry / light-visibility extraction (gen_geo.py's compute_geo / compute_vis) on the MI355X renderer.
compute_vis batches all (point, light) pairs and skips the colour net; it must equal the reference's scheme -- one
light at a time through the full render() -- bit for bit (rays are independent; weights never see the colour net)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_compute_geo_and_vis(tmp_path):
    from oracle import geo as og
    from tests.test_gpu_neus_render import _build
    from vqnerf_release_amd.geo.gen_geo import GeoExtractor, intersect_circle
    cfg, sdf, col, var, ren = _build('small')
    ren.n_importance, ren.up_sample_steps = 16, 4
    H = W = 12
    o, d, near, far = [torch.tensor(a).cuda() for a in og.make_rays(H * W, 3)]
    ex = GeoExtractor(ren, max_radius=2.0, light_h=16, max_rays=4096)
    geo = ex.compute_geo(o, d, near, far, perturb_overwrite=0)
    assert geo['surf'].shape == (H * W, 3) and set(np.unique(geo['mask'].cpu().numpy())) <= {0.0, 1.0}
    nn = geo['normal'].norm(dim=-1)
    assert torch.allclose(nn, torch.ones_like(nn), atol=1e-5)
    assert geo['mask'].sum() > 10                                   # the sphere-initialised SDF is hit
    lvis = ex.compute_vis(geo['surf'], geo['normal'], geo['mask'], perturb_overwrite=0)
    assert lvis.shape == (H * W, 512) and float(lvis.min()) >= -1e-6 and float(lvis.max()) <= 1 + 1e-6
    assert float(lvis[geo['mask'][:, 0] == 0].abs().max()) == 0.0
    # the reference's scheme: light by light, full render (colour net included), front-lit rays only
    fg = geo['mask'][:, 0] > 0
    pts, nrm = geo['surf'][fg], geo['normal'][fg]
    lx = ex.lxyz.cuda()
    ref = torch.zeros(pts.shape[0], 512, device='cuda')
    for i in range(0, 512, 37):                                      # a subset of the lights keeps the test short
        s2l = lx[:, i, :] - pts
        s2l = s2l / s2l.norm(dim=-1, keepdim=True)
        front = (s2l * nrm).sum(-1) > 0
        if not bool(front.any()):
            continue
        oo, dd = pts[front].contiguous(), s2l[front].contiguous()
        ff, _ = intersect_circle(oo, dd, 2.0)
        nnr = torch.minimum(torch.full_like(ff, 0.1), ff / 2.0)
        with torch.no_grad():
            r = ren.render(oo, dd, nnr, ff, 2.0, perturb_overwrite=0, cos_anneal_ratio=1.0, background_rgb=torch.ones(1, 3).cuda())
        ref[front, i] = 1.0 - r['weight_sum'][:, 0]
        assert torch.equal(lvis[fg][:, i], ref[:, i])
    ex.save_view(str(tmp_path / 'train_000'), H, W, geo, lvis)
    assert np.load(tmp_path / 'train_000' / 'lvis.npy').shape == (H, W, 512)
    assert np.load(tmp_path / 'train_000' / 'xyz.npy').dtype == np.float32


def test_compute_geo_and_vis_vs_oracle():
    """`GeoExtractor.compute_geo` / `compute_vis` (all (point, light) pairs of a chunk in one batch, colour net skipped) against
    the oracle's statement of gen_geo.py:182-344 (one `render` per light, lpix_chunk = 1, through the reference-pinned oracle
    renderer).  Both sides run without jitter (the reference draws a fresh one per call).  Tolerances: fp32 end-to-end render
    (1e-3 on weight sums / colours, see test_gpu_neus_render), 2e-3 on unit normals."""
    from oracle import geo as og, decomp as od
    from tests.test_gpu_neus_render import _build
    from vqnerf_release_amd.geo.gen_geo import GeoExtractor
    cfg, sdf, col, var, ren = _build('small')
    cfg = dict(cfg, renderer=dict(cfg['renderer'], n_importance=16))
    ren.n_importance, ren.up_sample_steps = 16, 4
    with torch.no_grad():
        var.variance.fill_(0.5)
    p_sdf, p_col = og.to_torch(og.make_sdf_params(cfg, 0)), og.to_torch(og.make_color_params(cfg, 1))
    R = 24
    o, d, near, far = [torch.tensor(a) for a in og.make_rays(R, seed=5, spread=0.12)]
    want = og.compute_geo(p_sdf, p_col, 0.5, cfg, o, d, near, far, 2.0, alpha_thres=0.5, batch_size=10)
    ex = GeoExtractor(ren, max_radius=2.0, light_h=16, max_rays=4096)
    got = ex.compute_geo(o.cuda(), d.cuda(), near.cuda(), far.cuda(), alpha_thres=0.5, perturb_overwrite=0)
    np.testing.assert_array_equal(got['mask'].cpu().numpy(), want['mask'])
    fg = want['mask'][:, 0] > 0
    assert 4 < fg.sum() < R                                            # hits and misses
    np.testing.assert_allclose(got['rgb'].cpu().numpy(), want['rgb'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(got['surf'].cpu().numpy(), want['surf'], rtol=0, atol=1e-3)
    np.testing.assert_allclose(got['normal'].cpu().numpy()[fg], want['normal'][fg], rtol=0, atol=2e-3)
    np.testing.assert_allclose(got['normal'].cpu().numpy()[~fg], 1 / np.sqrt(3), rtol=0, atol=1e-6)      # gen_geo.py:325-326
    # visibility from the SAME geometry buffers on both sides (isolates compute_vis)
    lxyz, _ = od.gen_light_xyz(16, 32)
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    want_l = og.compute_vis(p_sdf, p_col, 0.5, cfg, T(lxyz.reshape(1, -1, 3)), T(want['surf']), T(want['normal']), T(want['mask']), 2.0,
                            batch_size=7)
    got_l = ex.compute_vis(T(want['surf']).cuda(), T(want['normal']).cuda(), T(want['mask']).cuda(), perturb_overwrite=0).cpu().numpy()
    assert got_l.shape == want_l.shape == (R, 512)
    s2l = lxyz.reshape(1, -1, 3) - want['surf'][:, None, :]
    lcos = ((s2l / np.linalg.norm(s2l, axis=-1, keepdims=True)) * want['normal'][:, None, :]).sum(-1)
    sure = np.abs(lcos) > 1e-5                                          # front / back lit beyond rounding
    np.testing.assert_allclose(got_l[sure], want_l[sure], rtol=0, atol=1e-3)
    back = sure & (lcos <= 0)
    assert (got_l[back] == 0).all() and (want_l[back] == 0).all()      # back-lit lights are never traced (gen_geo.py:214-222)
    assert (want_l[~fg] == 0).all() and (got_l[~fg] == 0).all()
    assert 0.2 < (want_l[fg] != 0).mean() < 0.8 and (want_l[fg][want_l[fg] != 0] < 0.9).sum() > 20       # lit / unlit / partly occluded


def test_extract_views_shards_and_resumes(tmp_path):
    """The per-view loop of gen_geo.py:126-180 over a Blender-format set: view range split `num_p / p_i` (the reference's
    multi-GPU extraction), finished views skipped, files = the decomp loaders' contract."""
    import os
    from tests.test_datasets import _write_blender_set
    from tests.test_gpu_neus_render import _build
    from vqnerf_release_amd.geo import conf as hocon
    from vqnerf_release_amd.geo.gen_geo import GeoExtractor
    from vqnerf_release_amd.geo.models.nerfset import Dataset
    _write_blender_set(str(tmp_path / 'scene'), n=3, H=10, W=12)
    ds = Dataset(hocon.parse_string('dataset { data_dir = %s\n longint = false }' % (tmp_path / 'scene'))['dataset'], device='cuda')
    cfg, sdf, col, var, ren = _build('small')
    ren.n_importance, ren.up_sample_steps, ren.perturb = 16, 4, 0.0
    ex = GeoExtractor(ren, max_radius=ds.max_radius, light_h=16, max_rays=8192)
    out = str(tmp_path / 'surf')
    assert ex.extract_views(ds, out, is_train=True, num_p=2, p_i=1) == [2]          # ceil(3 / 2) = 2 views per shard
    assert ex.extract_views(ds, out, is_train=True, num_p=2, p_i=0) == [0, 1]
    assert sorted(os.listdir(out)) == ['train_000', 'train_001', 'train_002']
    assert set(os.listdir(os.path.join(out, 'train_001'))) == set(GeoExtractor.VIEW_FILES)
    lv = np.load(os.path.join(out, 'train_001', 'lvis.npy'))
    xyz, nrm = np.load(os.path.join(out, 'train_001', 'xyz.npy')), np.load(os.path.join(out, 'train_001', 'normal.npy'))
    assert lv.shape == (10, 12, 512) and xyz.shape == nrm.shape == (10, 12, 3) and lv.dtype == np.float32
    gt_bg = ds.masks[1, :, :, 0].cpu().numpy() == 0
    assert (lv[gt_bg] == 0).all()                                                    # visibility lives on the ground-truth mask
    assert ex.extract_views(ds, out, is_train=True) == []                            # everything finished: nothing to do
    os.remove(os.path.join(out, 'train_002', 'lvis.npy'))
    assert ex.extract_views(ds, out, is_train=True) == [2]
    assert ex.extract_views(ds, str(tmp_path / 'val'), is_train=False, no_vis=True, alpha_thres=0.8) == [0, 1, 2]
    assert 'lvis.npy' not in os.listdir(tmp_path / 'val' / 'val_000') and ex.extract_views(ds, str(tmp_path / 'val'), is_train=False, no_vis=True) == []
