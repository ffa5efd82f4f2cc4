"""Blender-format (NeRF-synthetic) image set for the NeuS trainer: mirror of geo/NeuS-ours2/models/nerfset.py.

Same constructor (`Dataset(conf, is_train=True)`, conf = the HOCON `dataset` block), attributes (`n_images, H, W, focal,
pose_all, images, masks, max_radius, object_bbox_min/max, near, far`) and methods (`gen_rays_at`, `gen_random_rays_at`,
`near_far_from_sphere`, `image_at`).  What is different, on purpose:

  * images, masks and poses live in HBM (an 800x800x100-view set is 1.5 GB of fp32, a rounding error of 288 GB), pixels
    are drawn with a device generator and gathered on the device: no per-step host gather + H2D copy
    (nerfset.py:113-130 indexes CPU tensors and uploads 10 floats per ray every step);
  * PNGs are decoded with Pillow (cv2 is not in this image).  The reference keeps cv2's channel order (B, G, R) for its
    colour targets (nerfset.py:44, :58 never swaps), so a colour network trained there emits BGR; `bgr=True` (default)
    keeps that order so checkpoints and rendered images stay interchangeable.  16-bit PNGs: both decoders keep the high
    byte (nerfset.py:153-156 `// 256`);
  * `new_h` resizing uses torch's bilinear kernel (cv2.resize's default, same half-pixel convention) on float data; the
    reference resizes the uint8 image, so values can differ by one 8-bit level.
"""
import json
import os
from glob import glob

import numpy as np
import torch
import torch.nn.functional as F


def _conf_get(conf, key, default):
    if conf is None:
        return default
    try:
        v = conf.get(key, default)
    except TypeError:
        v = conf.get(key)
    return default if v is None else v


def read_rgba_u8(path, bgr=True):
    """[H,W,4] uint8; 16-bit files keep the high byte.  bgr: cv2.imread(path, -1) channel order (B,G,R,A)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode in ('I;16', 'I;16B', 'I'):                         # 16-bit grey
        a = (np.asarray(im).astype(np.uint32) // 256).astype(np.uint8)
        a = np.stack([a, a, a, np.full_like(a, 255)], -1)
    else:
        a = np.asarray(im.convert('RGBA'))                        # Pillow reduces 16-bit RGB(A) to the high byte
    if bgr:
        a = a[..., [2, 1, 0, 3]]
    return np.ascontiguousarray(a)


class Dataset:
    def __init__(self, conf, is_train=True, device='cuda', bgr=True, seed=None):
        self.device = torch.device(device)
        self.conf = conf
        self.bgr = bgr
        self.data_dir = conf['data_dir'] if not hasattr(conf, 'get_string') else conf.get_string('data_dir')
        cams = 'transforms_train.json' if is_train else 'transforms_val.json'
        prefix = 'train_*' if is_train else 'val_*'
        self.render_cameras_name = self.object_cameras_name = cams
        self.camera_outside_sphere = bool(_conf_get(conf, 'camera_outside_sphere', True))
        self.near, self.far = float(_conf_get(conf, 'near', 2.0)), float(_conf_get(conf, 'far', 6.0))
        self.longint = bool(_conf_get(conf, 'longint', True))

        with open(os.path.join(self.data_dir, cams)) as f:
            self.camera_dict = json.load(f)
        self.images_lis = sorted(glob(os.path.join(self.data_dir, prefix)))
        self.n_images = len(self.images_lis)
        if self.n_images == 0:
            raise FileNotFoundError(f'no {prefix} view directories under {self.data_dir}')
        self.cx, self.cy = (self.camera_dict['cx'], self.camera_dict['cy']) if 'cx' in self.camera_dict else (None, None)

        rgba = np.stack([read_rgba_u8(os.path.join(d, 'rgba.png'), bgr=bgr) for d in self.images_lis]).astype(np.float32)
        new_h = float(_conf_get(conf, 'new_h', 0))
        if new_h > 0:
            h, w = rgba.shape[1:3]
            k = new_h / h
            t = torch.from_numpy(rgba).permute(0, 3, 1, 2)
            t = F.interpolate(t, size=(int(new_h), int(w * k)), mode='bilinear', align_corners=False)
            rgba = t.permute(0, 2, 3, 1).round().clamp(0, 255).numpy()
            if self.cx is not None:
                self.cx, self.cy = self.cx * k, self.cy * k
        self.images = torch.from_numpy(rgba[..., :3] / 255.0).float().to(self.device)                     # [n, H, W, 3]
        self.masks = torch.from_numpy(np.repeat(rgba[..., 3:], 3, -1) / 255.0).float().to(self.device)     # [n, H, W, 3]

        poses = []
        for idx in range(self.n_images):
            m = self.camera_dict['frames'][idx]['transform_matrix']
            if isinstance(m, str):
                m = [float(x) for x in m.split(',')]
            poses.append(np.asarray(m, np.float32).reshape(4, 4))
        self.pose_all = torch.from_numpy(np.stack(poses)).to(self.device)                                 # c2w [n, 4, 4]
        self.H, self.W = self.images.shape[1], self.images.shape[2]
        self.focal = 0.5 * self.W / np.tan(0.5 * self.camera_dict['camera_angle_x'])
        self.image_pixels = self.H * self.W
        self.max_radius = self._get_radius()
        self.object_bbox_min = np.array([-1.1, -1.1, -1.1]) * self.max_radius
        self.object_bbox_max = np.array([1.1, 1.1, 1.1]) * self.max_radius
        self.gen = torch.Generator(device=self.device)
        if seed is not None:
            self.gen.manual_seed(int(seed))

    def _centre(self):
        return (self.W // 2, self.H // 2) if self.cx is None else (int(self.cx), int(self.cy))

    def _dirs(self, img_idx, px, py):
        """Pixel coordinates (any shape) -> (rays_o, unit rays_v) in world space (nerfset.py:98-104, :122-129)."""
        cx, cy = self._centre()
        p = torch.stack([(px - cx) / self.focal, -(py - cy) / self.focal, -torch.ones_like(py)], -1)
        R, t = self.pose_all[img_idx, :3, :3], self.pose_all[img_idx, :3, 3]
        v = (R @ p[..., None]).squeeze(-1)
        v = v / torch.linalg.norm(v, ord=2, dim=-1, keepdim=True)
        return t.expand(v.shape), v

    def gen_rays_at(self, img_idx, resolution_level=1, gen_mask=False):
        """All rays of one camera, [H/l, W/l, 3] each (nerfset.py:86-108)."""
        l = resolution_level
        tx = torch.linspace(0, self.W - 1, self.W // l, device=self.device)
        ty = torch.linspace(0, self.H - 1, self.H // l, device=self.device)
        py, px = torch.meshgrid(ty, tx, indexing='ij')
        rays_o, rays_v = self._dirs(img_idx, px, py)
        if gen_mask:
            return rays_o, rays_v, self.masks[img_idx, :, :, :1]
        return rays_o, rays_v

    def gen_random_rays_at(self, img_idx, batch_size):
        """[batch_size, 10] = rays_o | rays_v | colour | mask, for uniformly drawn pixels of one image (nerfset.py:110-130)."""
        px = torch.randint(0, self.W, (batch_size,), device=self.device, generator=self.gen)
        py = torch.randint(0, self.H, (batch_size,), device=self.device, generator=self.gen)
        color = self.images[img_idx][py, px]
        mask = self.masks[img_idx][py, px]
        rays_o, rays_v = self._dirs(img_idx, px.float(), py.float())
        return torch.cat([rays_o, rays_v, color, mask[:, :1]], -1)

    def near_far_from_sphere(self, rays_o, rays_d):
        shape = rays_d.shape[:-1] + (1,)
        return (torch.full(shape, self.near, device=rays_d.device), torch.full(shape, self.far, device=rays_d.device))

    def _get_radius(self):
        """Largest distance from the origin of the near / far points on any camera's optical axis (nerfset.py:138-145)."""
        bd = np.array([[0.0, 0.0], [0.0, 0.0], [-self.near, -self.far], [1.0, 1.0]])
        r = 0.0
        for c2w in self.pose_all.cpu().numpy():
            r = max(r, float(np.max(np.sqrt(np.sum(np.square((c2w @ bd)[:3, :]), axis=0)))))
        return r

    def image_at(self, idx, resolution_level):
        """uint8 [H/l, W/l, 3] of view idx in the set's channel order (nerfset.py:147-150)."""
        img = self.images[idx].permute(2, 0, 1)[None] * 255.0
        if resolution_level != 1:
            img = F.interpolate(img, size=(self.H // resolution_level, self.W // resolution_level), mode='bilinear',
                                align_corners=False)
        return img[0].permute(1, 2, 0).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
