"""Per-view geometry + light-visibility extraction: mirror of `Runner.compute_geo` / `compute_vis` / `intersect_circle`
/ `normal_correct` of geo/NeuS-ours2/gen_geo.py (:182-369) on the MI355X renderer.  These produce the inputs of the
decomp half (`xyz.npy`, `normal.npy`, `alpha.png`, `rgb.png`, `lvis.npy` per view, gen_geo.py:329-342,256-257).

compute_vis is the heaviest consumer of `render()` in the whole pipeline: for every foreground pixel one
secondary ray per front-lit light (~256 of 512).  The reference walks the lights one at a time (`lpix_chunk=1`) with a
host sync per light; here all (point, light) pairs of a chunk go through the kernels in one batch, stay on the device,
and the colour network is skipped because only `weight_sum` is used (gen_geo.py:239-242).
"""
import os

import numpy as np
import torch

from vqnerf_release_amd.decomp.brdf.renderer import gen_light_xyz


def intersect_circle(x, d, r, eps=1e-7):
    """far intersection of rays x + t d with the sphere |p| = r (gen_geo.py:346-357)."""
    b = 2.0 * (x * d).sum(-1)
    a = (d * d).sum(-1)
    c = (x * x).sum(-1) - r ** 2
    denom = torch.where(2 * a > eps, 2 * a, torch.full_like(a, eps))
    disc = torch.sqrt(b * b - 4.0 * a * c)
    t = torch.maximum((-b + disc) / denom, (-b - disc) / denom)
    return t[:, None], x + t[:, None] * d


def normal_correct(rays_o, surf, normal):
    surf2c = rays_o - surf
    surf2c = surf2c / torch.linalg.norm(surf2c, dim=-1, keepdim=True)
    return torch.where((surf2c * normal).sum(-1, keepdim=True) >= 0, normal, -normal)


class GeoExtractor:
    def __init__(self, renderer, max_radius, use_white_bkgd=True, cos_anneal_ratio=1.0, light_h=16, max_rays=1 << 20):
        self.renderer, self.max_radius = renderer, float(max_radius)
        self.use_white_bkgd, self.car, self.max_rays = use_white_bkgd, cos_anneal_ratio, int(max_rays)
        lxyz, _ = gen_light_xyz(light_h, 2 * light_h)
        self.lxyz = torch.tensor(lxyz.reshape(1, -1, 3), dtype=torch.float32)

    @torch.no_grad()
    def compute_geo(self, rays_o, rays_d, near, far, alpha_thres=0.5, perturb_overwrite=-1):
        """rays [R,3] -> dict(rgb [R,3], surf [R,3], normal [R,3], mask [R,1]) on the device (gen_geo.py:259-308)."""
        bg = torch.ones(1, 3, device=rays_o.device) if self.use_white_bkgd else None
        out = {k: [] for k in ('rgb', 'surf', 'normal', 'mask')}
        for s in range(0, rays_o.shape[0], self.max_rays):
            o, d = rays_o[s:s + self.max_rays].contiguous(), rays_d[s:s + self.max_rays].contiguous()
            r = self.renderer.render(o, d, near[s:s + self.max_rays], far[s:s + self.max_rays], self.max_radius,
                                     perturb_overwrite=perturb_overwrite, cos_anneal_ratio=self.car, background_rgb=bg)
            nrm = (r['gradients'] * r['weights'][:, :, None] * r['inside_sphere'][..., None]).sum(1)
            nrm = nrm / torch.sqrt((nrm * nrm).sum(-1, keepdim=True))
            out['rgb'].append(r['color_fine'])
            out['surf'].append(r['surf'])
            out['normal'].append(normal_correct(o, r['surf'], nrm))
            out['mask'].append((r['weight_sum'] > alpha_thres).float())
        out = {k: torch.cat(v, 0) for k, v in out.items()}
        # background pixels get the unit diagonal normal (gen_geo.py:321-322)
        diag = torch.full_like(out['normal'], 1.0 / np.sqrt(3.0))
        out['normal'] = out['normal'] * out['mask'] + diag * (1.0 - out['mask'])
        return out

    @torch.no_grad()
    def compute_vis(self, surf, normal, mask, perturb_overwrite=-1):
        """surf, normal [R,3], mask [R,1] -> lvis [R, L] (zeros on background pixels and back-lit lights; gen_geo.py:182-257)."""
        dev = surf.device
        lxyz = self.lxyz.to(dev)
        L = lxyz.shape[1]
        fg = mask[:, 0] > 0
        pts, nrm = surf[fg], normal[fg]
        M = pts.shape[0]
        lvis_fg = torch.zeros(M, L, device=dev)
        bg = torch.ones(1, 3, device=dev) if self.use_white_bkgd else None
        ren = self.renderer
        prev, ren.weights_only = ren.weights_only, True
        try:
            step = max(1, self.max_rays // L)
            for s in range(0, M, step):
                p, n = pts[s:s + step], nrm[s:s + step]
                surf2l = lxyz - p[:, None, :]
                surf2l = surf2l / torch.linalg.norm(surf2l, dim=-1, keepdim=True)
                front = torch.einsum('ijk,ik->ij', surf2l, n) > 0
                if not bool(front.any()):
                    continue
                pi, li = front.nonzero(as_tuple=True)
                o, d = p[pi].contiguous(), surf2l[pi, li].contiguous()
                far, _ = intersect_circle(o, d, self.max_radius)
                near = torch.minimum(torch.full_like(far, 0.1), far / 2.0)
                r = ren.render(o, d, near, far, self.max_radius, perturb_overwrite=perturb_overwrite, cos_anneal_ratio=self.car,
                               background_rgb=bg)
                lvis_fg[s + pi, li] = 1.0 - r['weight_sum'][:, 0]
        finally:
            ren.weights_only = prev
        lvis = torch.zeros(surf.shape[0], L, device=dev)
        lvis[fg] = lvis_fg
        return lvis

    @staticmethod
    def save_view(view_dir, H, W, geo, lvis=None):
        """The on-disk contract of the decomp loaders (shape_unit.py:68-78): xyz.npy, normal.npy [H,W,3], lvis.npy [H,W,L]
        as float32 np.save; rgb.png / alpha.png / normal.png / lvis.png 8-bit previews (PIL; BGR order as cv2.imwrite)."""
        os.makedirs(view_dir, exist_ok=True)
        npf = lambda t, c: t.detach().float().cpu().numpy().reshape(H, W, c)
        np.save(os.path.join(view_dir, 'xyz.npy'), npf(geo['surf'], 3))
        np.save(os.path.join(view_dir, 'normal.npy'), npf(geo['normal'], 3))
        if lvis is not None:
            np.save(os.path.join(view_dir, 'lvis.npy'), npf(lvis, lvis.shape[-1]))
        try:
            from PIL import Image
        except ImportError:
            return
        u8 = lambda a: np.clip(a, 0, 255).astype(np.uint8)
        Image.fromarray(u8(npf(geo['rgb'], 3) * 256)).save(os.path.join(view_dir, 'rgb.png'))
        Image.fromarray(u8(npf(geo['mask'], 1)[..., 0] * 256)).save(os.path.join(view_dir, 'alpha.png'))
        Image.fromarray(u8(npf(geo['normal'], 3) * 128 + 128)).save(os.path.join(view_dir, 'normal.png'))
        if lvis is not None:
            Image.fromarray(u8(npf(lvis, lvis.shape[-1]).mean(-1) * 256)).save(os.path.join(view_dir, 'lvis.png'))
        Image.fromarray(u8(npf(geo['surf'], 3))).save(os.path.join(view_dir, 'xyz.png'))    # the reference's clipped preview

    VIEW_FILES = ('lvis.npy', 'lvis.png', 'alpha.png', 'normal.npy', 'normal.png', 'rgb.png', 'xyz.npy', 'xyz.png')

    @classmethod
    def check_finished(cls, view_dir, no_vis=False):
        """All files of a view present (gen_geo.py:371-381): lets an interrupted / sharded run pick up where it stopped."""
        return all(os.path.exists(os.path.join(view_dir, f)) for f in cls.VIEW_FILES if not (no_vis and f.startswith('lvis')))

    @torch.no_grad()
    def extract_views(self, dataset, scene_out_dir, is_train=True, resolution_level=1, num_p=None, p_i=None, no_vis=False,
                      alpha_thres=0.5, log=None):
        """The per-view loop of gen_geo.py:126-180: geometry buffers (+ light visibility unless `no_vis`) of every view of
        `dataset` into `scene_out_dir/{train,val}_{idx:03d}/`.  Views are independent, so multi-GPU extraction is a split of
        the view range with no collective -- `--num_p / --p_i` in the reference (README.md:45-53), by default rank /
        world_size of the process group here.  Training views use the ground-truth mask for visibility (the decomp stage
        trains on it) and alpha_thres 0.5; validation views use the predicted mask at `alpha_thres`.  Returns the indices done."""
        import math
        from vqnerf_release_amd import parallel
        if num_p is None:
            num_p, p_i = parallel.world_size(), parallel.rank()
        n = dataset.n_images
        p_step = math.ceil(n / num_p)
        prefix = 'train_' if is_train else 'val_'
        done = []
        for idx in range(p_i * p_step, min(n, (p_i + 1) * p_step)):
            view_dir = os.path.join(scene_out_dir, prefix + '{i:03d}'.format(i=idx))
            if self.check_finished(view_dir, no_vis=no_vis):
                continue
            rays = dataset.gen_rays_at(idx, resolution_level=resolution_level)
            rays_o, rays_d = rays[0], rays[1]
            H, W, _ = rays_o.shape
            o, d = rays_o.reshape(-1, 3).contiguous(), rays_d.reshape(-1, 3).contiguous()
            near, far = dataset.near_far_from_sphere(o, d)
            geo = self.compute_geo(o, d, near, far, alpha_thres=0.5 if is_train else alpha_thres)
            lvis = None
            if not no_vis:
                mask = geo['mask']
                gt = getattr(dataset, 'masks', None)
                if is_train and gt is not None and tuple(gt.shape[1:3]) == (H, W):
                    mask = (gt[idx, :, :, :1].reshape(-1, 1) > 0).float().to(o.device)
                lvis = self.compute_vis(geo['surf'], geo['normal'], mask)
            self.save_view(view_dir, H, W, geo, lvis)
            done.append(idx)
            if log is not None:
                log(f'{prefix}{idx:03d}: done')
        return done
