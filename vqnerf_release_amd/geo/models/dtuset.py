"""Projection-matrix image set (DTU / hand-held captures) for the NeuS trainer: mirror of geo/NeuS-ours2/models/dtuset.py.

On disk (kept): `<data_dir>/{train,val}.json` with per-view `world_mat` (K [R|t], 4x4) and `scale_mat` (unit-sphere
normalisation, 4x4) lists (dtuset.py:20-33, :54-58) and one `train_*/rgba.png` / `val_*/rgba.png` directory per view.
Same constructor, attributes (`n_images, H, W, pose_all, intrinsics_all, intrinsics_all_inv, scale_mats_np, images, masks,
max_radius = 1, near, far, object_bbox_min/max`) and methods as the reference class.  As in models/nerfset.py of this
package, images / masks / cameras are resident in HBM, pixels are drawn and gathered on the device, PNGs are decoded by
Pillow in cv2's channel order, and P = K [R | t] is factored by an RQ decomposition (cv2.decomposeProjectionMatrix in the
reference, dtuset.py:163-181).
"""
import json
import os
from glob import glob

import numpy as np
import torch
import torch.nn.functional as F

from vqnerf_release_amd.geo.models.nerfset import _conf_get, read_rgba_u8


def decompose_projection_matrix(P):
    """P [3,4] = K [R | -R c] -> (4x4 intrinsics with K[2,2] = 1, 4x4 camera-to-world pose) (dtuset.py:163-181)."""
    from scipy.linalg import rq
    K, R = rq(P[:3, :3])
    S = np.diag(np.sign(np.diag(K)))                 # K with a positive diagonal
    K, R = K @ S, S @ R
    if np.linalg.det(R) < 0:
        R = -R
    c = -np.linalg.solve(P[:3, :3], P[:3, 3])
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K / K[2, 2]
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.transpose()
    pose[:3, 3] = c
    return intrinsics, pose


class Dataset:
    def __init__(self, conf, is_train=True, device='cuda', bgr=True, seed=None):
        self.device = torch.device(device)
        self.conf = conf
        self.bgr = bgr
        self.data_dir = conf['data_dir'] if not hasattr(conf, 'get_string') else conf.get_string('data_dir')
        cams = 'train.json' if is_train else 'val.json'
        prefix = 'train_*' if is_train else 'val_*'
        self.render_cameras_name = self.object_cameras_name = cams
        self.camera_outside_sphere = bool(_conf_get(conf, 'camera_outside_sphere', True))
        with open(os.path.join(self.data_dir, cams)) as f:
            self.camera_dict = json.load(f)
        self.images_lis = sorted(glob(os.path.join(self.data_dir, prefix)))
        self.n_images = len(self.images_lis)
        if self.n_images == 0:
            raise FileNotFoundError(f'no {prefix} view directories under {self.data_dir}')

        rgba = np.stack([read_rgba_u8(os.path.join(d, 'rgba.png'), bgr=bgr) for d in self.images_lis]).astype(np.float32)
        new_h = float(_conf_get(conf, 'new_h', 0))
        self.k = 1.0
        if new_h > 0:
            h, w = rgba.shape[1:3]
            self.k = new_h / h
            t = torch.from_numpy(rgba).permute(0, 3, 1, 2)
            t = F.interpolate(t, size=(int(new_h), int(w * self.k)), mode='bilinear', align_corners=False)
            rgba = t.permute(0, 2, 3, 1).round().clamp(0, 255).numpy()
        self.images = torch.from_numpy(rgba[..., :3] / 255.0).float().to(self.device)                     # [n, H, W, 3]
        self.masks = torch.from_numpy(np.repeat(rgba[..., 3:], 3, -1) / 255.0).float().to(self.device)     # [n, H, W, 3]
        self.H, self.W = self.images.shape[1], self.images.shape[2]
        self.image_pixels = self.H * self.W

        poses, intr, self.scale_mats_np = [], [], []
        for idx in range(self.n_images):
            raw_scale = np.array(self.camera_dict['scale_mat'][idx], np.float64)
            raw_world = np.array(self.camera_dict['world_mat'][idx], np.float64)
            intrinsic, pose = decompose_projection_matrix((raw_world @ raw_scale)[0:3, 0:4])
            intrinsic[:2, :3] = intrinsic[:2, :3] * self.k
            self.scale_mats_np.append(raw_scale.astype(np.float32))
            poses.append(pose.astype(np.float32))
            intr.append(intrinsic.astype(np.float32))
        self.pose_all = torch.from_numpy(np.stack(poses)).to(self.device)                                  # c2w [n, 4, 4]
        self.intrinsics_all = torch.from_numpy(np.stack(intr)).to(self.device)
        self.intrinsics_all_inv = torch.from_numpy(np.linalg.inv(np.stack(intr).astype(np.float64)).astype(np.float32)).to(self.device)

        self.max_radius = 1.0
        self.near, self.far = self.compute_near_far()
        eps = 0.01
        bmin = np.array([-(self.max_radius + eps)] * 3 + [1.0])
        bmax = np.array([self.max_radius + eps] * 3 + [1.0])
        inv0 = np.linalg.inv(self.scale_mats_np[0])
        self.object_bbox_min = (inv0 @ self.scale_mats_np[0] @ bmin[:, None])[:3, 0]
        self.object_bbox_max = (inv0 @ self.scale_mats_np[0] @ bmax[:, None])[:3, 0]
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(int(seed))

    def _dirs(self, img_idx, px, py):
        """Pixel coordinates (any shape) -> (rays_o, unit rays_v) in world space (dtuset.py:100-105, :116-121)."""
        p = torch.stack([px, py, torch.ones_like(py)], -1)
        p = (self.intrinsics_all_inv[img_idx, :3, :3] @ p[..., None]).squeeze(-1)
        v = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
        v = (self.pose_all[img_idx, :3, :3] @ v[..., None]).squeeze(-1)
        return self.pose_all[img_idx, :3, 3].expand(v.shape), v

    def gen_rays_at(self, img_idx, resolution_level=1):
        """All rays of one camera, [H/l, W/l, 3] each (dtuset.py:92-106)."""
        l = resolution_level
        tx = torch.linspace(0, self.W - 1, self.W // l, device=self.device)
        ty = torch.linspace(0, self.H - 1, self.H // l, device=self.device)
        py, px = torch.meshgrid(ty, tx, indexing='ij')
        return self._dirs(img_idx, px, py)

    def gen_random_rays_at(self, img_idx, batch_size):
        """[batch_size, 10] = rays_o | rays_v | colour | mask for uniformly drawn pixels of one image (dtuset.py:108-122)."""
        px = torch.randint(0, self.W, (batch_size,), device=self.device, generator=self.gen)
        py = torch.randint(0, self.H, (batch_size,), device=self.device, generator=self.gen)
        color = self.images[img_idx][py, px]
        mask = self.masks[img_idx][py, px]
        rays_o, rays_v = self._dirs(img_idx, px.float(), py.float())
        return torch.cat([rays_o, rays_v, color, mask[:, :1]], -1)

    def compute_near_far(self):
        """Camera-space depth of the two points where the line camera -> origin meets the bounding sphere, min / max over
        the views (dtuset.py:124-140)."""
        nears, fars = [], []
        for pose in self.pose_all.cpu().numpy().astype(np.float64):
            cam = pose[:3, 3:]
            n_p = cam / np.linalg.norm(cam, ord=2, axis=0, keepdims=True) * self.max_radius
            w2c = np.linalg.inv(pose)
            nears.append((w2c @ np.concatenate([n_p, [[1.0]]], 0))[2, 0])
            fars.append((w2c @ np.concatenate([-n_p, [[1.0]]], 0))[2, 0])
        return float(np.min(nears)), float(np.max(fars))

    def near_far_from_sphere(self, rays_o, rays_d):
        """Per ray: the parameter of the point closest to the origin -/+ 1 (dtuset.py:142-149)."""
        a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
        b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return mid - 1.0, mid + 1.0

    def image_at(self, idx, resolution_level):
        """uint8 [H/l, W/l, 3] of view idx in the set's channel order (dtuset.py:151-154)."""
        img = self.images[idx].permute(2, 0, 1)[None] * 255.0
        if resolution_level != 1:
            img = F.interpolate(img, size=(self.H // resolution_level, self.W // resolution_level), mode='bilinear',
                                align_corners=False)
        return img[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
