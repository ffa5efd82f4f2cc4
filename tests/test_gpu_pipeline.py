"""GPU: the two halves hand over through files, as in the reference's scripts -- image set -> NeuS trainer -> per-view
geometry / visibility buffers (gen_geo) -> reflectance loader -> VQ-stage training -> inference output.  Checks that the
file contracts line up end to end on a tiny scene (nothing here is a quality claim)."""
import json
import os
import re

import numpy as np
import pytest
import torch

from tests.decomp_util import make_config

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_image_set_to_relit_views(tmp_path):
    from tests.test_datasets import _write_blender_set, _pose
    from vqnerf_release_amd.geo.nerf_runner import Runner
    from vqnerf_release_amd.geo.gen_geo import GeoExtractor
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.datasets import get_dataset_class
    from vqnerf_release_amd.decomp.nerfactor.util import io as ioutil
    H, W, n = 24, 32, 3
    scene = tmp_path / 'scene'
    scene.mkdir()
    _write_blender_set(str(scene), n=n, H=H, W=W)
    for i in range(n):                                               # NeRFactor's per-view camera file next to rgba.png
        m = _pose(0.7 * i)
        json.dump({'imh': H, 'imw': W, 'cam_angle_x': 0.6911, 'cam_transform_mat': ','.join(repr(float(v)) for v in m.reshape(-1))},
                  open(scene / ('train_%03d' % i) / 'metadata.json', 'w'))
    # 1. geometry stage: a few optimisation steps, then the per-view buffers
    text = open(os.path.join(HERE, 'golden', 'neus_like.conf')).read().replace('./exp/', str(tmp_path) + '/exp/')
    text = text.replace('warm_up_end = 5000', 'warm_up_end = 0').replace('batch_size = 64', 'batch_size = 128')
    text = re.sub(r'data_dir = [^\n]*', 'data_dir = %s/\n    longint = false' % scene, text, count=1)
    torch.manual_seed(0)
    r = Runner(conf_text=text, case='toy')
    r.update_learning_rate()
    for it in range(3):
        r.train_step(r.dataset.gen_random_rays_at(it % n, r.batch_size))
    r.renderer.perturb = 0.0
    ex = GeoExtractor(r.renderer, max_radius=r.dataset.max_radius, light_h=16, max_rays=1 << 16)
    assert ex.extract_views(r.dataset, str(tmp_path / 'surf'), is_train=True) == [0, 1, 2]
    # 2. reflectance stage reads them
    cfg = make_config(data_root=str(scene), data_nerf_root=str(tmp_path / 'surf'), imh=H, use_nerf_alpha='False', cache='True',
                      n_rays_per_step=32, num_embed=5, num_drop=1, thres_str='0.3', epochs=2, ckpt_period=2, vali_period=0,
                      total_sample_vq=32, random_seed=1, cluster_center_path='')
    ds = get_dataset_class('shape_unit')(cfg, 'train', device='cuda')
    assert ds.get_n_views() == n and not ds.incomplete_paths
    view = ds.view(1)
    assert view[0] == ['train_001'] and view[7].shape == (H * W, 3) and view[9].shape == (H * W, 512)
    xyz_file = np.load(tmp_path / 'surf' / 'train_001' / 'xyz.npy').reshape(-1, 3)
    fg = view[5][:, 0].cpu().numpy() > 0
    moved = np.linalg.norm(xyz_file - view[2].cpu().numpy(), axis=-1) == 0             # the loader nudges camera-collapsed points
    np.testing.assert_allclose(view[7].cpu().numpy()[fg & ~moved], xyz_file[fg & ~moved], atol=1e-6)
    rays_o, _ = r.dataset.gen_rays_at(1)
    np.testing.assert_allclose(view[2].cpu().numpy(), rays_o.reshape(-1, 3).cpu().numpy(), atol=1e-5)   # both halves agree on the camera
    if fg.sum() < 40:
        pytest.skip('random image set left too little foreground for the pair sampler')
    model, hist = train_nfr.fit(cfg, str(tmp_path / 'vq'), ds, None, log=lambda *_: None)
    assert len(hist['loss']) == 2 and all(np.isfinite(hist['loss'])) and os.path.exists(tmp_path / 'vq' / 'checkpoints' / 'ckpt-2.pt')
    # 3. inference: every view relit under two probes
    os.makedirs(tmp_path / 'probes')
    rng = np.random.default_rng(0)
    for name in ('a', 'b'):
        ioutil.write_hdr(str(tmp_path / 'probes' / (name + '.hdr')), rng.uniform(0, 2, (16, 32, 3)).astype(np.float32))
    model.config.set('DEFAULT', 'test_envmap_dir', str(tmp_path / 'probes'))
    model._novel_lights()
    model.to('cuda')
    w, k = train_nfr.render_views(model, ds, str(tmp_path / 'relit'), relight_probes=True)
    w.flush()
    assert k == n
    for i in range(n):
        files = set(os.listdir(tmp_path / 'relit' / ('batch%09d' % i)))
        assert {'pred_rgb_probes_a.png', 'pred_rgb_probes_b.png', 'pred_albedo.png', 'pred_rough.npy', 'metadata.json'} <= files
