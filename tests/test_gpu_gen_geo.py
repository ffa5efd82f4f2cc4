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
