"""Geo trainer: mirror of the training half of geo/NeuS-ours2/nerf_runner.py (`Runner.__init__` :22-97,
`train` :99-175, cosine LR with warm-up :186-195, cos-anneal :180-184, checkpoint keys :210-232) on the MI355X
classes, plus rank-sharded data parallelism (the reference trains on one GPU).

Out of scope here: TensorBoard, mesh export.  The dataset is any object with the reference's
dataset contract -- `n_images`, `max_radius`, `gen_random_rays_at(img_idx, batch_size) -> [B,10]` (o, d, rgb, mask) and
`near_far_from_sphere(rays_o, rays_d)`: `models.nerfset.Dataset` when the conf's `dataset.data_dir` holds a Blender-format
image set (as nerf_runner.py:36 does), `models.dtuset.Dataset` for a world_mat / scale_mat set (dtu_runner.py:36), `SyntheticDataset` (tests, bench) otherwise.
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from vqnerf_release_amd import parallel
from vqnerf_release_amd.geo import conf as hocon
from vqnerf_release_amd.geo.models.fields import RenderingNetwork, SDFNetwork, SingleVarianceNetwork, NeRF
from vqnerf_release_amd.geo.models.renderer import NeuSRenderer


class SyntheticDataset:
    """Stand-in for models/nerfset.py: one pin-hole camera per image on a circle of radius 4 looking at the origin,
    constant near/far (nerfset.py:132-136), colours and masks from a fixed analytic pattern.  Rays are generated on
    the device (no host gather / H2D copy per step)."""

    def __init__(self, conf=None, n_images=8, H=800, W=800, near=2.0, far=6.0, device='cuda', seed=0):
        conf = conf or {}
        self.n_images, self.H, self.W = int(conf.get('n_train', n_images)), H, W
        self.near, self.far = float(conf.get('near', near)), float(conf.get('far', far))
        self.device = torch.device(device)
        self.focal = 0.5 * W / math.tan(0.5 * 0.6911)
        ang = torch.arange(self.n_images, dtype=torch.float32) * (2 * math.pi / max(self.n_images, 1))
        self.cam_o = torch.stack([4 * torch.sin(ang), torch.zeros_like(ang), 4 * torch.cos(ang)], -1).to(self.device)
        self.max_radius = 2.0
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def _frame(self, idx):
        o = self.cam_o[idx]
        fwd = -o / o.norm()
        up = torch.tensor([0.0, 1.0, 0.0], device=self.device)
        right = torch.linalg.cross(fwd, up)
        right = right / right.norm()
        return o, fwd, right, torch.linalg.cross(right, fwd)

    def gen_random_rays_at(self, img_idx, batch_size):
        idx = int(img_idx) % self.n_images
        px = torch.randint(0, self.W, (batch_size,), generator=self.gen, device=self.device).float()
        py = torch.randint(0, self.H, (batch_size,), generator=self.gen, device=self.device).float()
        o, fwd, right, up = self._frame(idx)
        d = (fwd[None] + ((px - 0.5 * self.W + 0.5) / self.focal)[:, None] * right[None]
             - ((py - 0.5 * self.H + 0.5) / self.focal)[:, None] * up[None])
        d = d / d.norm(dim=-1, keepdim=True)
        # analytic target: a unit-ish ball with a smooth colour field; mask = ray hits the 0.6-ball
        b = (o[None] * d).sum(-1)
        hit = (b * b - (o.dot(o) - 0.36)) > 0
        rgb = 0.5 + 0.5 * torch.sin(torch.stack([px, py, px + py], -1) * 0.01)
        rgb = torch.where(hit[:, None], rgb, torch.ones_like(rgb))
        return torch.cat([o[None].expand(batch_size, 3), d, rgb, hit[:, None].float()], -1)

    def near_far_from_sphere(self, rays_o, rays_d):
        n = rays_o.shape[0]
        return (torch.full((n, 1), self.near, device=rays_o.device), torch.full((n, 1), self.far, device=rays_o.device))


class Runner:
    GRAPH_WARMUP = 2

    def __init__(self, conf_path=None, mode='train', case='CASE_NAME', is_continue=False, conf_text=None, dataset=None,
                 device='cuda', graph=False):
        """`graph=True`: the optimisation step -- up-sampling passes, forward / backward tile programs, compositing, weight-gradient
        contractions, weight-norm chain rule, Adam -- is captured once into a HIP graph (after GRAPH_WARMUP eager steps) and
        replayed on static buffers: ~150 of its ~260 launches are a few microseconds long and the host cannot issue them as fast as
        the device retires them.  Conditions (checked): fixed batch size, constant cos_anneal_ratio (annealing finished or off),
        `perturb` draws from the device generator (they do).  The learning rate is a device scalar refreshed before every replay."""
        self.device = torch.device(device)
        self.graph = bool(graph)
        self._cap = None
        if conf_text is None:
            with open(conf_path) as f:
                conf_text = f.read()
        self.conf_path = conf_path
        self.conf = hocon.parse_string(conf_text.replace('CASE_NAME', case))
        self.base_exp_dir = self.conf['general.base_exp_dir']
        if dataset is None:
            dconf = self.conf.get('dataset')
            data_dir = dconf.get('data_dir') if dconf is not None else None
            if data_dir and os.path.isfile(os.path.join(data_dir, 'transforms_train.json')):
                from vqnerf_release_amd.geo.models.nerfset import Dataset
                dataset = Dataset(dconf, is_train=(mode == 'train'), device=self.device)
            elif data_dir and os.path.isfile(os.path.join(data_dir, 'train.json')):     # world_mat / scale_mat sets (dtu_runner.py:36)
                from vqnerf_release_amd.geo.models.dtuset import Dataset
                dataset = Dataset(dconf, is_train=(mode == 'train'), device=self.device)
            else:
                dataset = SyntheticDataset(dconf, device=self.device)
        self.dataset = dataset
        # data parallel: every rank must draw its OWN pixels (an unseeded device generator starts from the same state everywhere)
        if parallel.world_size() > 1 and hasattr(dataset, 'gen'):
            dataset.gen.manual_seed(dataset.gen.initial_seed() + 7919 * parallel.rank())
        self.iter_step = 0
        t = self.conf['train']
        self.end_iter, self.save_freq, self.report_freq = t.get_int('end_iter'), t.get_int('save_freq'), t.get_int('report_freq')
        self.val_freq, self.val_mesh_freq = t.get_int('val_freq'), t.get_int('val_mesh_freq')
        self.batch_size = t.get_int('batch_size')
        self.lr_end_iter = t.get_int('lr_end_iter', -1)                 # dtu_runner.py:41, :192-194
        self.validate_resolution_level = t.get_int('validate_resolution_level')
        self.learning_rate, self.learning_rate_alpha = t.get_float('learning_rate'), t.get_float('learning_rate_alpha')
        self.use_white_bkgd = t.get_bool('use_white_bkgd')
        self.warm_up_end, self.anneal_end = t.get_float('warm_up_end', 0.0), t.get_float('anneal_end', 0.0)
        self.igr_weight, self.mask_weight = t.get_float('igr_weight'), t.get_float('mask_weight')
        self.is_continue, self.mode = is_continue, mode

        m = self.conf['model']
        self.nerf_outside = NeRF(**m['nerf']).to(self.device)
        self.sdf_network = SDFNetwork(**m['sdf_network']).to(self.device)
        self.deviation_network = SingleVarianceNetwork(**m['variance_network']).to(self.device)
        self.color_network = RenderingNetwork(**m['rendering_network']).to(self.device)
        params = (list(self.nerf_outside.parameters()) + list(self.sdf_network.parameters())
                  + list(self.deviation_network.parameters()) + list(self.color_network.parameters()))
        if self.graph:
            # step counters and lr on the device (the update lives inside the captured step): one HIP launch on a GPU (optim.HipAdam)
            lr_t = torch.tensor(float(self.learning_rate), device=self.device)
            if torch.device(self.device).type == 'cuda':
                from vqnerf_release_amd.optim import HipAdam
                self.optimizer = HipAdam(params, lr=lr_t)
            else:
                self.optimizer = torch.optim.Adam(params, lr=lr_t, capturable=True, fused=True)
        else:
            self.optimizer = torch.optim.Adam(params, lr=self.learning_rate)
        self.renderer = NeuSRenderer(self.nerf_outside, self.sdf_network, self.deviation_network, self.color_network,
                                     **m['neus_renderer'])
        # data parallel: one flat bucket [grads of the nets that are evaluated || loss terms] -> one all-reduce / step
        self._dp_params = [p for net in (self.sdf_network, self.deviation_network, self.color_network) for p in net.parameters()]
        self.bucket = None
        self.last_stats = {}
        if is_continue:
            names = sorted(n for n in os.listdir(os.path.join(self.base_exp_dir, 'checkpoints'))
                           if n.endswith('pth') and int(n[5:-4]) <= self.end_iter)
            if names:
                self.load_checkpoint(names[-1])

    # ---- schedules (nerf_runner.py:180-195) ----
    def get_cos_anneal_ratio(self):
        return 1.0 if self.anneal_end == 0.0 else float(np.min([1.0, self.iter_step / self.anneal_end]))

    def update_learning_rate(self):
        if self.iter_step < self.warm_up_end:
            factor = self.iter_step / self.warm_up_end
        else:
            end = self.end_iter if self.lr_end_iter < 0 else self.lr_end_iter
            progress = (self.iter_step - self.warm_up_end) / (end - self.warm_up_end)
            factor = (np.cos(np.pi * progress) + 1.0) * 0.5 * (1 - self.learning_rate_alpha) + self.learning_rate_alpha
        for g in self.optimizer.param_groups:
            if torch.is_tensor(g['lr']):
                g['lr'].fill_(float(self.learning_rate * factor))         # (graph mode: a device scalar the captured Adam reads)
            else:
                g['lr'] = float(self.learning_rate * factor)

    def get_image_perm(self):
        return torch.randperm(self.dataset.n_images)

    # ---- one optimisation step (the loop body of nerf_runner.py:105-147) ----
    def train_step(self, data, t_rand=None):
        """data [B,10] = o, d, rgb, mask: this rank's rays.  Under data parallelism the loss is normalised by the GLOBAL
        mask sum / ray count, gradients are summed over ranks in one bucket, and every rank takes the same step."""
        if self.graph and t_rand is None and self.get_cos_anneal_ratio() == 1.0 and self.iter_step >= self.GRAPH_WARMUP:
            return self._train_step_replay(data)
        stats = self._train_step_body(data, t_rand)
        self._finish_step(stats)
        return self.last_stats

    def _finish_step(self, extra):
        self.iter_step += 1
        self.update_learning_rate()
        self.last_stats = {'loss': extra[0], 'color_loss': extra[1], 'eikonal_loss': extra[2], 'mask_loss': extra[3]}

    def _train_step_replay(self, data):
        if self._cap is None:
            self._static_data = data.clone()
            cap = parallel.SegmentedCapture()
            with cap:
                self._static_extra = self._train_step_body(self._static_data, None, fresh_leaves=True)
            self._cap = cap
        if data.shape != self._static_data.shape:
            raise ValueError(f'the captured step takes batches of shape {tuple(self._static_data.shape)}, got {tuple(data.shape)}')
        self._static_data.copy_(data)
        self._cap.replay()
        import vqnerf_release_amd
        vqnerf_release_amd.weights_changed()              # a replay moves the weights without bumping any tensor `_version`
        self._finish_step(self._static_extra)
        return self.last_stats

    def _train_step_body(self, data, t_rand=None, fresh_leaves=False):
        rays_o, rays_d, true_rgb, mask = data[:, :3], data[:, 3:6], data[:, 6:9], data[:, 9:10]
        near, far = self.dataset.near_far_from_sphere(rays_o, rays_d)
        bg = torch.ones([1, 3], device=data.device) if self.use_white_bkgd else None
        mask = (mask > 0.5).float() if self.mask_weight > 0.0 else torch.ones_like(mask)
        world = parallel.world_size()
        sums = torch.stack([mask.sum(), torch.full((), float(mask.numel()), device=data.device)])
        if world > 1:
            parallel.all_reduce_sum(sums, what='all_reduce:loss_normalisers')
        mask_sum, n_rays = sums[0] + 1e-5, sums[1]
        if self.bucket is None:
            self.bucket = parallel.FlatBucket(self._dp_params, n_extra=4)
        self.optimizer.zero_grad(set_to_none=True)

        def loss_terms():
            out = self.renderer.render(rays_o, rays_d, near, far, self.dataset.max_radius, background_rgb=bg,
                                       cos_anneal_ratio=self.get_cos_anneal_ratio(), t_rand=t_rand)
            color_error = (out['color_fine'] - true_rgb) * mask
            color_loss = color_error.abs().sum() / mask_sum
            # eikonal: the renderer returns the mean over this rank's samples; weight by the rank's share of rays
            share = mask.numel() / n_rays
            eik = out['gradient_error'] * share
            mask_loss = F.binary_cross_entropy(out['weight_sum'].clip(1e-3, 1.0 - 1e-3), mask, reduction='sum') / n_rays
            return color_loss + eik * self.igr_weight + mask_loss * self.mask_weight, color_loss, eik, mask_loss

        if fresh_leaves:
            # (capture only) the step runs on fresh leaf aliases of the parameters: a parameter's AccumulateGrad node is bound to the
            # stream it was first used on, and the autograd engine would pull that (default) stream into the capture -- see
            # decomp/nerfactor/train_nfr.py, Trainer._step
            from contextlib import ExitStack
            from torch.nn.utils.stateless import _reparametrize_module
            leaves = [p.detach().requires_grad_(True) for p in self.bucket.params]
            by_id = {id(p): q for p, q in zip(self.bucket.params, leaves)}
            with ExitStack() as st:
                for net in (self.sdf_network, self.deviation_network, self.color_network):
                    st.enter_context(_reparametrize_module(net, {n: by_id[id(p)] for n, p in net.named_parameters() if id(p) in by_id}))
                loss, color_loss, eik, mask_loss = loss_terms()
                grads = torch.autograd.grad(loss, leaves, allow_unused=True)
            with torch.no_grad():
                have = [(v, g) for v, g in zip(self.bucket.views, grads) if g is not None]
                for p, v, g in zip(self.bucket.params, self.bucket.views, grads):
                    if g is None:
                        v.zero_()
                    p.grad = v
                if have:
                    parallel.multi_copy([v for v, _ in have], [g for _, g in have])      # one launch, not one per parameter
        else:
            self.bucket.attach()
            loss, color_loss, eik, mask_loss = loss_terms()
            loss.backward()
        with torch.no_grad():
            ex = self.bucket.extra
            ex[0], ex[1], ex[2], ex[3] = loss.detach(), color_loss.detach(), eik.detach(), mask_loss.detach()
        extra = self.bucket.all_reduce()
        self.optimizer.step()
        return extra

    def train(self, n_iters=None, log=None):
        self.update_learning_rate()
        res_step = self.end_iter - self.iter_step if n_iters is None else n_iters
        perm = self.get_image_perm()
        for _ in range(res_step):
            data = self.dataset.gen_random_rays_at(perm[self.iter_step % len(perm)], self.batch_size)
            stats = self.train_step(data)
            if self.iter_step % self.report_freq == 0 and parallel.rank() == 0:      # one host sync per report, not per step
                msg = 'iter:{:8>d} loss = {:.6f} lr={}'.format(self.iter_step, float(stats['loss']), self.optimizer.param_groups[0]['lr'])
                (log or print)(msg)
            if self.iter_step % self.save_freq == 0 and parallel.rank() == 0:
                self.save_checkpoint()
            if self.val_freq > 0 and self.iter_step % self.val_freq == 0 and parallel.rank() == 0 and hasattr(self.dataset, 'gen_rays_at'):
                self.validate_image()                         # nerf_runner.py:152-153
            if self.iter_step % len(perm) == 0:
                perm = self.get_image_perm()

    # ---- validation images (nerf_runner.py:236-343) ----
    @torch.no_grad()
    def validate_image(self, idx=-1, resolution_level=-1, is_train=True, max_rays=1 << 18):
        """Render view `idx` at 1/resolution_level and write what the reference writes: `validations_fine/` (prediction
        stacked over the input image), `alpha/` (weight_sum > 0.5), `inside_sphere/`, `normals/` (unit normals over a
        (1,1,1)/sqrt(3) background, * 128 + 128), named `{iter:08d}_0_{idx}.png` under base_exp_dir (`test/` below it when
        not is_train).  Rays are rendered `max_rays` at a time by the fused kernels and the images are composed on the
        device: one copy per image.  Arrays are handed to the PNG encoder in cv2's channel convention (B, G, R), as the
        reference does with cv.imwrite.  Returns {name: uint8 array as written}."""
        from PIL import Image
        if idx < 0:
            idx = int(np.random.randint(self.dataset.n_images))
        if resolution_level < 0:
            resolution_level = self.validate_resolution_level
        rays_o, rays_d = self.dataset.gen_rays_at(idx, resolution_level=resolution_level)[:2]
        H, W, _ = rays_o.shape
        rays_o, rays_d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
        bg = torch.ones([1, 3], device=rays_o.device) if self.use_white_bkgd else None
        n_s = self.renderer.n_samples + self.renderer.n_importance
        rgb, mask, inside, normals = [], [], [], []
        for o, d in zip(rays_o.split(max_rays), rays_d.split(max_rays)):
            o, d = o.contiguous(), d.contiguous()
            near, far = self.dataset.near_far_from_sphere(o, d)
            out = self.renderer.render(o, d, near, far, self.dataset.max_radius, cos_anneal_ratio=self.get_cos_anneal_ratio(),
                                       background_rgb=bg)
            rgb.append(out['color_fine'])
            mask.append((out['weight_sum'] > 0.5).float())
            w = out['weights'][:, :n_s, None] * out['inside_sphere'][..., None]
            normals.append((out['gradients'] * w).sum(dim=1))
            inside.append(w.sum(dim=1))
        rgb, mask, inside, normals = torch.cat(rgb), torch.cat(mask), torch.cat(inside), torch.cat(normals)

        def unit(v):                                             # _np_norm (nerf_runner.py:397-403)
            r = v.norm(dim=-1, keepdim=True)
            return torch.where(r == 0, torch.full_like(v, math.sqrt(1.0 / 3.0)), v / r)

        nrm = unit(normals) * mask + unit(torch.ones_like(normals)) * (1.0 - mask)
        u8 = lambda t: t.clip(0, 255).to(torch.uint8).cpu().numpy()
        imgs = {'validations_fine': u8(rgb.reshape(H, W, 3) * 256), 'alpha': u8(mask.reshape(H, W) * 256),
                'inside_sphere': u8(inside.reshape(H, W) * 256), 'normals': u8(nrm.reshape(H, W, 3) * 128 + 128)}
        if is_train:
            imgs['validations_fine'] = np.concatenate([imgs['validations_fine'], self.dataset.image_at(idx, resolution_level)])
        base = self.base_exp_dir if is_train else os.path.join(self.base_exp_dir, 'test')
        for name, arr in imgs.items():
            os.makedirs(os.path.join(base, name), exist_ok=True)
            a = arr[..., ::-1] if arr.ndim == 3 else arr         # (B, G, R) array -> RGB file, what cv.imwrite does
            Image.fromarray(np.ascontiguousarray(a)).save(os.path.join(base, name, '{:0>8d}_{}_{}.png'.format(self.iter_step, 0, idx)))
        return imgs

    # ---- checkpoints: same keys / file names as nerf_runner.py:210-232 ----
    def save_checkpoint(self):
        ckpt = {'nerf': self.nerf_outside.state_dict(), 'sdf_network_fine': self.sdf_network.state_dict(),
                'variance_network_fine': self.deviation_network.state_dict(),
                'color_network_fine': self.color_network.state_dict(), 'optimizer': self.optimizer.state_dict(),
                'iter_step': self.iter_step}
        d = os.path.join(self.base_exp_dir, 'checkpoints')
        os.makedirs(d, exist_ok=True)
        torch.save(ckpt, os.path.join(d, 'ckpt_{:0>6d}.pth'.format(self.iter_step)))

    def load_checkpoint(self, checkpoint_name):
        ckpt = torch.load(os.path.join(self.base_exp_dir, 'checkpoints', checkpoint_name), map_location=self.device, weights_only=False)
        self.nerf_outside.load_state_dict(ckpt['nerf'])
        self.sdf_network.load_state_dict(ckpt['sdf_network_fine'])
        self.deviation_network.load_state_dict(ckpt['variance_network_fine'])
        self.color_network.load_state_dict(ckpt['color_network_fine'])
        self.optimizer.load_state_dict(ckpt['optimizer'])
        self.iter_step = ckpt['iter_step']
