"""CPU: the two on-disk data paths into the kernels -- Blender-format image sets for the NeuS trainer
(geo/models/nerfset.py) and per-view geometry buffers for the reflectance stages (datasets/shape_unit.py) -- against files
written here in the reference's layouts, incl. the round trip gen_geo.save_view -> shape_unit.Dataset -> outer_sample."""
import json
import math
import os

import numpy as np
import pytest
import torch

from tests.decomp_util import make_config

PIL = pytest.importorskip('PIL.Image')


def _pose(angle, dist=4.0):
    """Blender-style c2w looking at the origin from a circle in the xz-plane (camera looks down its -z)."""
    o = np.array([dist * math.sin(angle), 0.0, dist * math.cos(angle)])
    z = o / np.linalg.norm(o)                     # camera +z points away from the scene
    x = np.cross([0.0, 1.0, 0.0], z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, o
    return m


def _write_blender_set(root, n=3, H=12, W=16, as_str=False, seed=0):
    rng = np.random.default_rng(seed)
    frames, imgs = [], []
    for i in range(n):
        d = os.path.join(root, 'train_%03d' % i)
        os.makedirs(d)
        im = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        PIL.fromarray(im, 'RGBA').save(os.path.join(d, 'rgba.png'))
        imgs.append(im)
        m = _pose(0.7 * i)
        frames.append({'transform_matrix': ','.join(repr(float(v)) for v in m.reshape(-1)) if as_str else m.tolist()})
    with open(os.path.join(root, 'transforms_train.json'), 'w') as f:
        json.dump({'camera_angle_x': 0.6911, 'frames': frames}, f)
    return np.stack(imgs)


@pytest.mark.parametrize('as_str', [False, True])
def test_nerfset_matches_the_files(tmp_path, as_str):
    from vqnerf_release_amd.geo import conf as hocon
    from vqnerf_release_amd.geo.models.nerfset import Dataset
    imgs = _write_blender_set(str(tmp_path), as_str=as_str)
    conf = hocon.parse_string('dataset { data_dir = %s\n longint = false }' % tmp_path)['dataset']
    ds = Dataset(conf, is_train=True, device='cpu', seed=3)
    n, H, W = imgs.shape[:3]
    assert (ds.n_images, ds.H, ds.W) == (n, H, W)
    # colour targets keep cv2's channel order (B,G,R); mask = alpha repeated
    np.testing.assert_allclose(ds.images.numpy(), imgs[..., [2, 1, 0]] / 255.0, atol=1e-7)
    np.testing.assert_allclose(ds.masks.numpy(), np.repeat(imgs[..., 3:], 3, -1) / 255.0, atol=1e-7)
    assert abs(ds.focal - 0.5 * W / math.tan(0.5 * 0.6911)) < 1e-9
    assert abs(ds.max_radius - 2.0) < 1e-5                          # cameras at distance 4, near 2 / far 6
    np.testing.assert_allclose(ds.object_bbox_max, 1.1 * ds.max_radius)
    # full-image rays: unit, origin = camera centre, the principal-point pixel looks down the camera's -z
    o, v = ds.gen_rays_at(1)
    assert o.shape == v.shape == (H, W, 3)
    np.testing.assert_allclose(v.norm(dim=-1).numpy(), 1.0, atol=1e-6)
    c2w = ds.pose_all[1].numpy()
    np.testing.assert_allclose(o[0, 0].numpy(), c2w[:3, 3], atol=1e-6)
    np.testing.assert_allclose(v[H // 2, W // 2].numpy(), -c2w[:3, 2], atol=1e-6)
    o2, v2, m2 = ds.gen_rays_at(1, resolution_level=2, gen_mask=True)
    assert v2.shape == (H // 2, W // 2, 3) and m2.shape == (H, W, 1)
    # random rays: each row's colour / mask is the pixel its direction passes through
    rays = ds.gen_random_rays_at(2, 200)
    assert rays.shape == (200, 10)
    R = ds.pose_all[2, :3, :3]
    p = rays[:, 3:6] @ R                                             # R^T d
    px = torch.round(p[:, 0] / -p[:, 2] * ds.focal + W // 2).long()
    py = torch.round(-p[:, 1] / -p[:, 2] * ds.focal + H // 2).long()
    assert px.min() >= 0 and px.max() < W and py.min() >= 0 and py.max() < H and len(set(px.tolist())) > 4
    assert torch.equal(rays[:, 6:9], ds.images[2][py, px]) and torch.equal(rays[:, 9], ds.masks[2][py, px][:, 0])
    near, far = ds.near_far_from_sphere(rays[:, :3], rays[:, 3:6])
    assert near.shape == (200, 1) and float(near[0]) == 2.0 and float(far[0]) == 6.0
    assert ds.image_at(0, 1).dtype == np.uint8 and ds.image_at(0, 2).shape == (H // 2, W // 2, 3)
    np.testing.assert_array_equal(ds.image_at(0, 1), imgs[0][..., [2, 1, 0]])


def test_nerfset_new_h_resizes_images_and_principal_point(tmp_path):
    from vqnerf_release_amd.geo import conf as hocon
    from vqnerf_release_amd.geo.models.nerfset import Dataset
    _write_blender_set(str(tmp_path), H=12, W=16)
    meta = json.load(open(tmp_path / 'transforms_train.json'))
    meta.update(cx=8.0, cy=6.0)
    json.dump(meta, open(tmp_path / 'transforms_train.json', 'w'))
    conf = hocon.parse_string('dataset { data_dir = %s\n longint = false\n new_h = 6 }' % tmp_path)['dataset']
    ds = Dataset(conf, device='cpu')
    assert (ds.H, ds.W) == (6, 8) and (ds.cx, ds.cy) == (4.0, 3.0)
    assert float(ds.images.min()) >= 0 and float(ds.images.max()) <= 1
    with pytest.raises(FileNotFoundError):
        Dataset(conf, is_train=False, device='cpu')                 # no transforms_val.json / val_* views


def _write_decomp_view(data_root, nerf_root, vid, H, W, L, rng, collapse=True):
    from vqnerf_release_amd.geo.gen_geo import GeoExtractor
    c2w = _pose(0.3)
    os.makedirs(os.path.join(data_root, vid))
    rgba = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
    rgba[:, :W // 4, 3] = 0                                          # a background band
    rgba[:, W // 4:, 3] = 255
    PIL.fromarray(rgba, 'RGBA').save(os.path.join(data_root, vid, 'rgba.png'))
    with open(os.path.join(data_root, vid, 'metadata.json'), 'w') as f:
        json.dump({'imh': H, 'imw': W, 'cam_angle_x': 0.6911, 'cam_transform_mat': ','.join(repr(float(v)) for v in c2w.reshape(-1))}, f)
    surf = rng.uniform(-1, 1, (H * W, 3)).astype(np.float32)
    nrm = rng.normal(size=(H * W, 3)).astype(np.float32) * 3.0       # not unit: the loader re-normalises
    if collapse:
        surf[0] = c2w[:3, 3]                                         # occupancy 0: "surface" on the camera
        nrm[1] = 0.0
    lvis = rng.uniform(-0.2, 1.2, (H * W, L)).astype(np.float32)     # the loader clips to [0,1]
    geo = {'surf': torch.tensor(surf), 'normal': torch.tensor(nrm), 'rgb': torch.rand(H * W, 3),
           'mask': torch.tensor((rgba[..., 3].reshape(-1, 1) > 0).astype(np.float32))}
    GeoExtractor.save_view(os.path.join(nerf_root, vid), H, W, geo, lvis=torch.tensor(lvis))
    return rgba, surf, nrm, lvis, c2w


def _decomp_cfg(tmp_path, **over):
    return make_config(data_root=str(tmp_path / 'data'), data_nerf_root=str(tmp_path / 'geo'), imh=over.pop('imh', 16),
                       use_nerf_alpha='False', cache='True', **over)


def test_shape_unit_round_trip_from_save_view(tmp_path):
    from vqnerf_release_amd.decomp.nerfactor.datasets import get_dataset_class
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    rng = np.random.default_rng(0)
    H, W, L = 16, 20, 512
    made = {vid: _write_decomp_view(str(tmp_path / 'data'), str(tmp_path / 'geo'), vid, H, W, L, rng) for vid in ('train_000', 'train_001')}
    os.makedirs(tmp_path / 'data' / 'train_002')                     # a view without buffers is skipped, not fatal
    json.dump({}, open(tmp_path / 'data' / 'train_002' / 'metadata.json', 'w'))
    cfg = _decomp_cfg(tmp_path, n_rays_per_step=32)
    ds = get_dataset_class('shape_unit')(cfg, 'train', device='cpu')
    assert ds.get_n_views() == 2 and len(ds.incomplete_paths) == 1 and ds.bs == 32
    id_, hw, rayo, rayd, rgb, alpha, pred_alpha, xyz, normal, lvis = ds.view(0)
    rgba, surf, nrm, lv, c2w = made['train_000']
    N = H * W
    assert id_ == ['train_000'] and hw.shape == (N, 2) and hw[0].tolist() == [H, W]
    assert rayo.shape == rayd.shape == rgb.shape == xyz.shape == normal.shape == (N, 3)
    assert alpha.shape == pred_alpha.shape == (N, 1) and lvis.shape == (N, L)
    # rays: pixel (y, x) -> ((x - W/2)/f, -(y - H/2)/f, -1) rotated to the world, not normalised (shape_unit.py:286-291)
    f = 0.5 * W / math.tan(0.5 * 0.6911)
    y, x = 5, 7
    want = c2w[:3, :3] @ np.array([(x - 0.5 * W) / f, -(y - 0.5 * H) / f, -1.0])
    np.testing.assert_allclose(rayd[y * W + x].numpy(), want, atol=1e-6)
    np.testing.assert_allclose(rayo[3].numpy(), c2w[:3, 3], atol=1e-6)
    # rgb composited on white with the image's own alpha; alpha from rgba, pred_alpha from the geometry stage
    a = rgba[..., 3:].reshape(N, 1) / 255.0
    np.testing.assert_allclose(alpha.numpy(), a, atol=1e-6)
    np.testing.assert_allclose(rgb.numpy(), rgba[..., :3].reshape(N, 3) / 255.0 * a + (1 - a), atol=1e-6)
    assert set(np.unique(pred_alpha.numpy())) <= {0.0, 1.0}
    # buffers: unit normals, zero normal -> +y, collapsed xyz pushed 0.1 along its ray, visibility clipped
    np.testing.assert_allclose(normal.norm(dim=-1).numpy(), 1.0, atol=1e-5)
    np.testing.assert_allclose(normal[1].numpy(), [0, 1, 0], atol=1e-7)
    np.testing.assert_allclose(normal[2].numpy(), nrm[2] / np.linalg.norm(nrm[2]), atol=1e-6)
    np.testing.assert_allclose(xyz[0].numpy(), (rayo[0] + 0.1 * rayd[0]).numpy(), atol=1e-6)
    np.testing.assert_array_equal(xyz[5].numpy(), surf[5])
    np.testing.assert_array_equal(lvis.numpy(), np.clip(lv, 0, 1))
    assert ds.view(0) is ds.view(0)                                  # cached hand-off
    # an epoch visits every view once; the pair sampler runs on what the loader yields
    views = list(ds.build_pipeline(seed=1))
    assert sorted(v[0][0] for v in views) == ['train_000', 'train_001']
    out = train_nfr.outer_sample(views[0], cfg, 'nerf', generator=torch.Generator().manual_seed(0))
    assert out[0] == views[0][0] and out[7].shape == (64, 3) and out[9].shape == (64, L)
    assert (out[5] > 0.9).all()                                      # both pixels of each pair are foreground
    # validation mode: batch size = pixels of a view
    os.rename(tmp_path / 'data' / 'train_000', tmp_path / 'data' / 'val_000')
    os.rename(tmp_path / 'geo' / 'train_000', tmp_path / 'geo' / 'val_000')
    dv = get_dataset_class('shape_unit')(cfg, 'vali', device='cpu')
    assert dv.get_n_views() == 1 and dv.bs == N


def test_shape_unit_resizes_to_imh(tmp_path):
    from vqnerf_release_amd.decomp.nerfactor.datasets.shape_unit import Dataset, resize_hw
    rng = np.random.default_rng(1)
    H, W, L = 16, 24, 8
    rgba, surf, nrm, lv, c2w = _write_decomp_view(str(tmp_path / 'data'), str(tmp_path / 'geo'), 'train_000', H, W, L, rng, collapse=False)
    ds = Dataset(_decomp_cfg(tmp_path, imh=8), 'train', device='cpu')
    b = ds.view(0)
    assert b[1][0].tolist() == [8, 12] and b[7].shape == (96, 3) and b[9].shape == (96, L)
    # shrinking by an integer factor = box average (cv2.INTER_AREA)
    want = np.clip(lv, None, None).reshape(8, 2, 12, 2, L).mean((1, 3))
    np.testing.assert_allclose(resize_hw(lv.reshape(H, W, L), 8), want, atol=1e-6)
    np.testing.assert_allclose(b[9].numpy(), np.clip(want, 0, 1).reshape(96, L), atol=1e-6)
    up = resize_hw(np.arange(12, dtype=np.float32).reshape(3, 4), 6)
    assert up.shape == (6, 8) and up.min() >= 0 and up.max() <= 11


def test_dtu_projection_decomposition():
    from vqnerf_release_amd.decomp.nerfactor.datasets.shape_unit import Dataset
    K = np.array([[900.0, 0.5, 320.0], [0, 880.0, 240.0], [0, 0, 1.0]])
    c2w = _pose(1.1, dist=3.0)
    R, c = c2w[:3, :3].T, c2w[:3, 3]
    P = K @ np.concatenate([R, (-R @ c)[:, None]], 1)
    for scale in (1.0, -2.5):                                        # a projection matrix is defined up to scale
        intr, pose = Dataset.decompose_projection_matrix(scale * P)
        np.testing.assert_allclose(intr[:3, :3], K, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(pose[:3, :3], c2w[:3, :3], atol=1e-6)
        np.testing.assert_allclose(pose[:3, 3], c, atol=1e-5)


def _cv_pose(angle, dist=2.5):
    """OpenCV-style c2w (x right, y down, z forward) looking at the origin."""
    o = np.array([dist * math.sin(angle), 0.3, dist * math.cos(angle)])
    z = -o / np.linalg.norm(o)
    x = np.cross(z, [0.0, -1.0, 0.0]); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = x, y, z, o
    return m


def _write_dtu_set(root, n=3, H=10, W=14, seed=5):
    """train.json with world_mat / scale_mat lists + train_*/rgba.png; returns (images, unit-frame c2w poses, K)."""
    rng = np.random.default_rng(seed)
    K = np.array([[30.0 * W / 14, 0.0, W / 2.0, 0], [0, 28.0 * H / 10, H / 2.0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1.0]])
    scale = np.diag([2.0, 2.0, 2.0, 1.0]); scale[:3, 3] = [0.1, -0.2, 0.3]      # unit sphere -> world
    world, imgs, poses = [], [], []
    for i in range(n):
        d = os.path.join(str(root), 'train_%03d' % i)
        os.makedirs(d)
        im = rng.integers(0, 256, (H, W, 4), dtype=np.uint8)
        PIL.fromarray(im, 'RGBA').save(os.path.join(d, 'rgba.png'))
        imgs.append(im)
        c2w = _cv_pose(0.9 * i)                                    # pose in the normalised (unit-sphere) frame
        poses.append(c2w)
        world.append((K @ np.linalg.inv(c2w) @ np.linalg.inv(scale)).tolist())    # so that world @ scale = K [R|t]
    with open(os.path.join(str(root), 'train.json'), 'w') as f:
        json.dump({'world_mat': world, 'scale_mat': [scale.tolist()] * n}, f)
    return np.stack(imgs), poses, K


def test_dtuset_matches_the_files(tmp_path):
    """models/dtuset.py contract (dtuset.py:12-154): world_mat @ scale_mat is factored into K and the pose, rays come from
    K^-1 and the pose, near / far from the unit sphere."""
    from vqnerf_release_amd.geo import conf as hocon
    from vqnerf_release_amd.geo.models.dtuset import Dataset
    n, H, W = 3, 10, 14
    imgs, poses, K = _write_dtu_set(tmp_path, n, H, W)
    conf = hocon.parse_string('dataset { data_dir = %s }' % tmp_path)['dataset']
    ds = Dataset(conf, device='cpu', seed=1)
    assert (ds.n_images, ds.H, ds.W, ds.max_radius) == (n, H, W, 1.0)
    np.testing.assert_allclose(ds.images.numpy(), imgs[..., [2, 1, 0]] / 255.0, atol=1e-7)
    np.testing.assert_allclose(ds.pose_all.numpy(), np.stack(poses), atol=1e-5)
    np.testing.assert_allclose(ds.intrinsics_all[0].numpy(), K, atol=1e-4)
    np.testing.assert_allclose(ds.object_bbox_max, [1.01] * 3, atol=1e-6)
    # cameras sit at |c| = sqrt(2.5^2 + 0.3^2) looking at the origin: depth of the sphere's near / far points
    dist = math.sqrt(2.5 ** 2 + 0.3 ** 2)
    assert abs(ds.near - (dist - 1.0)) < 1e-5 and abs(ds.far - (dist + 1.0)) < 1e-5
    o, v = ds.gen_rays_at(1)
    assert o.shape == v.shape == (H, W, 3)
    np.testing.assert_allclose(v.norm(dim=-1).numpy(), 1.0, atol=1e-6)
    np.testing.assert_allclose(v[5, 7].numpy(), poses[1][:3, 2], atol=1e-5)       # the principal-point pixel looks down +z
    rays = ds.gen_random_rays_at(2, 150)
    p = (rays[:, 3:6] @ ds.pose_all[2, :3, :3]).double()                          # R^T d, camera frame
    px = torch.round(p[:, 0] / p[:, 2] * 30.0 + 7.0).long()
    py = torch.round(p[:, 1] / p[:, 2] * 28.0 + 5.0).long()
    assert torch.equal(rays[:, 6:9], ds.images[2][py, px]) and torch.equal(rays[:, 9], ds.masks[2][py, px][:, 0])
    near, far = ds.near_far_from_sphere(rays[:, :3], rays[:, 3:6])
    mid = -(rays[:, :3] * rays[:, 3:6]).sum(-1, keepdim=True)
    np.testing.assert_allclose(near.numpy(), (mid - 1).numpy(), atol=1e-6)
    np.testing.assert_allclose((far - near).numpy(), 2.0, atol=1e-6)
    # new_h scales the intrinsics with the image (dtuset.py:37-45, :60)
    ds2 = Dataset(hocon.parse_string('dataset { data_dir = %s\n new_h = 20 }' % tmp_path)['dataset'], device='cpu')
    assert (ds2.H, ds2.W) == (20, 28)
    np.testing.assert_allclose(ds2.intrinsics_all[0, :2, :3].numpy(), 2 * K[:2, :3], atol=1e-3)
    assert ds.image_at(0, 2).shape == (H // 2, W // 2, 3)


def test_ref_nfr_dataset_adds_the_reference_colour_column(tmp_path):
    """Stage-3 loader (datasets/ref_nfr.py of the reference: :64-68, :258-273, :299-301): the shape_unit view + `ref` = rgb.png of the view's
    geometry directory (normalised 8-bit), between `normal` and `lvis`; the pair sampler carries the column along."""
    from PIL import Image
    from vqnerf_release_amd.decomp.nerfactor.datasets import get_dataset_class
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    rng = np.random.default_rng(4)
    H, W, L = 16, 20, 512
    for vid in ('train_000', 'train_001'):
        _write_decomp_view(str(tmp_path / 'data'), str(tmp_path / 'geo'), vid, H, W, L, rng, collapse=False)
    cfg = _decomp_cfg(tmp_path, n_rays_per_step=16, model='ref_nfr')
    ds = get_dataset_class('ref_nfr')(cfg, 'train', device='cpu')
    base = get_dataset_class('shape_unit')(cfg, 'train', device='cpu')
    v, b = ds.view(0), base.view(0)
    assert len(v) == 11 and len(b) == 10
    for i in range(2, 9):
        assert torch.equal(v[i], b[i])
    assert torch.equal(v[10], b[9])                                                  # lvis stays last
    png = np.asarray(Image.open(tmp_path / 'geo' / 'train_000' / 'rgb.png')).astype(np.float32) / 255.0
    np.testing.assert_allclose(v[9].numpy(), png.reshape(H * W, 3), rtol=0, atol=1e-7)
    os.remove(tmp_path / 'geo' / 'train_001' / 'rgb.png')                            # a view without its stage-2 render is skipped
    ds2 = get_dataset_class('ref_nfr')(cfg, 'train', device='cpu')
    assert ds2.get_n_views() == 1 and len(ds2.incomplete_paths) == 1
    batch = train_nfr.outer_sample(v, cfg, 'nerf', generator=torch.Generator().manual_seed(0), neighbour='max_diff')
    assert len(batch) == 11 and batch[9].shape == (32, 3) and batch[10].shape == (32, L)
    # `ref` rows are the gathered pixels' reference colours: find each sampled pixel through its (unique) ray direction
    d_all = v[3]
    for r in range(0, 32, 7):
        j = int(torch.nonzero((d_all == batch[3][r]).all(1))[0, 0])
        assert torch.equal(batch[9][r], v[9][j])
