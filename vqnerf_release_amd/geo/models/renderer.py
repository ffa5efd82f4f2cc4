"""NeuSRenderer on MI355X: host-side mirror of geo/NeuS-ours2/models/renderer.py:72-408 (same
constructor, same `render(...)` keyword arguments, same 11 result keys, `extract_geometry`).

Render / inference (no graph needed) is HIP end to end:
    vqn_neus_sdf_points  -> coarse SDF (renderer.py:337-338) and the SDF of each batch of new samples (:185)
    vqn_neus_upsample    -> up_sample + sample_pdf(det=True) (:131-175, :39-69)
    vqn_neus_merge       -> cat_z_vals (:177-191)
    vqn_neus_section_mids, vqn_neus_fine_points, vqn_neus_composite_fwd -> render_core (:193-297)
Training (a graph is needed, incl. the second-order eikonal term) keeps the no_grad up-sampling on those kernels and runs
render_core as explicit forward / backward tile programs (`vqn_tile_program`, `vqn_wgrad_partials`, geo/train_programs.py)
plus `vqn_neus_composite_fwd/_bwd` under two `torch.autograd.Function`s.  Networks whose shape the tile programs do not cover
take the torch-autograd statement of render_core instead -- loudly: a RuntimeWarning names the reason and
`renderer.last_train_backend` records which path the last graph-building render_core took ('hip' | 'torch').

The one random draw of the reference (`torch.rand([B,1]) - 0.5`, renderer.py:318) can be injected as
`t_rand` so that fixtures and data-parallel ranks are reproducible.
"""
import warnings

import numpy as np
import torch
import torch.nn.functional as F

from vqnerf_release_amd import _C
from vqnerf_release_amd.geo import packing
from vqnerf_release_amd.geo.models.fields import _needs_graph


class CompositeFunction(torch.autograd.Function):
    """renderer.py:229-282 as one differentiable op: vqn_neus_composite_fwd / vqn_neus_composite_bwd.
    Differentiable inputs: sdf [B,n], grad [B,n,3], rgb [B,n,3], inv_s [1]; differentiable outputs: color, weight_sum,
    gradient_error, weights.  The other outputs are detached statistics."""

    @staticmethod
    def forward(ctx, sdf, grad, rgb, inv_s, rays_o, rays_d, mid_z, dists, bg, radius, car):
        sdf, grad, rgb = sdf.detach().contiguous(), grad.detach().contiguous(), rgb.detach().contiguous()
        inv_s = inv_s.detach().reshape(1).contiguous()
        o = _C.neus_composite_fwd(rays_o, rays_d, mid_z, dists, sdf, grad, rgb, inv_s, bg, radius, car)
        gsum = o['gerr'].sum(0)
        gerr = gsum[0] / (gsum[1] + 1e-5)
        ctx.save_for_backward(sdf, grad, rgb, inv_s, rays_o, rays_d, mid_z, dists, gsum[1:2].contiguous())
        ctx.bg, ctx.radius, ctx.car = bg, radius, car
        ctx.mark_non_differentiable(o['cdf'], o['inside_sphere'], o['surf'], o['depth'], o['weight_max'])
        return o['color'], o['weight_sum'], gerr, o['weights'], o['cdf'], o['inside_sphere'], o['surf'], o['depth'], o['weight_max']

    @staticmethod
    def backward(ctx, g_color, g_wsum, g_gerr, g_weights, *unused):
        sdf, grad, rgb, inv_s, rays_o, rays_d, mid_z, dists, den = ctx.saved_tensors
        B, n = mid_z.shape
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=mid_z.device)
        g_color = z(B, 3) if g_color is None else g_color.contiguous()
        gw = None if g_wsum is None else g_wsum.reshape(B).contiguous()
        gwt = None if g_weights is None else g_weights.contiguous()
        gg = None if g_gerr is None else g_gerr.reshape(1).contiguous()
        g_sdf, g_grad, g_rgb, g_is = _C.neus_composite_bwd(rays_o, rays_d, mid_z, dists, sdf, grad, rgb, inv_s, ctx.bg, ctx.radius,
                                                           ctx.car, g_color, gw, gwt, gg, den)
        return g_sdf, g_grad, g_rgb, g_is.sum().reshape(1), None, None, None, None, None, None, None


def extract_fields(bound_min, bound_max, resolution, query_func):
    N = 64
    X = torch.linspace(bound_min[0], bound_max[0], resolution).split(N)
    Y = torch.linspace(bound_min[1], bound_max[1], resolution).split(N)
    Z = torch.linspace(bound_min[2], bound_max[2], resolution).split(N)
    u = np.zeros([resolution, resolution, resolution], dtype=np.float32)
    with torch.no_grad():
        for xi, xs in enumerate(X):
            for yi, ys in enumerate(Y):
                for zi, zs in enumerate(Z):
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing='ij')
                    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1).to(bound_min.device)
                    val = query_func(pts).reshape(len(xs), len(ys), len(zs)).detach().cpu().numpy()
                    u[xi * N: xi * N + len(xs), yi * N: yi * N + len(ys), zi * N: zi * N + len(zs)] = val
    return u


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func):
    import mcubes  # offline mesh export only (not on the hot path); same dependency as the reference
    u = extract_fields(bound_min, bound_max, resolution, query_func)
    vertices, triangles = mcubes.marching_cubes(u, threshold)
    b_max, b_min = bound_max.detach().cpu().numpy(), bound_min.detach().cpu().numpy()
    vertices = vertices / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :]
    return vertices, triangles


def sample_pdf(bins, weights, n_samples, det=False, u=None):
    """torch statement of renderer.py:39-69 (used by the autograd path and for non-deterministic u)."""
    weights = weights + 1e-5
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, steps=n_samples, device=bins.device)
            u = u.expand(list(cdf.shape[:-1]) + [n_samples])
        else:
            u = torch.rand(list(cdf.shape[:-1]) + [n_samples], device=bins.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bin_b + (u - cdf_b) / denom * (bin_a - bin_b)


class NeuSRenderer:
    def __init__(self, nerf, sdf_network, deviation_network, color_network, n_samples, n_importance, n_outside,
                 up_sample_steps, perturb):
        self.nerf = nerf
        self.sdf_network = sdf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb
        self._u = {}
        self._engines = {}
        self._step_pack = None            # x3 packs of the training render in flight (NeusTrainEngine.prepare_step)
        self.weights_only = False         # True: no-graph renders skip the colour net (weights / weight_sum / surf stay exact)
        self.matrix_mode = 'f32'          # 'f16s': no-graph renders on the split-precision kernels (f16 hi/lo MFMA; ~1e-6 relative, opt-in)
        self.train_backend = 'hip'        # 'hip': tile programs of geo/train_programs.py; 'torch': autograd over torch ops
        self.last_train_backend = None    # path of the last graph-building render_core: 'hip' (tile programs) | 'torch'

    # ---- packs shared by all kernels --------------------------------------------------------
    def _packs(self):
        mt, mode = self.color_network.max_tiles(), self.matrix_mode
        wb_s, d_s = self.sdf_network.packs(max_tiles=mt, mode=mode)
        wb_c, d_c = self.color_network.packs(feat_tiles=self.sdf_network.plan(mode=mode).tiles[-1], mode=mode)
        return wb_s, d_s, wb_c, d_c

    def _quantiles(self, m, device):
        key = (m, str(device))
        if key not in self._u:
            self._u[key] = torch.linspace(0.5 / m, 1.0 - 0.5 / m, steps=m, device=device).contiguous()
        return self._u[key]

    # ---- reference API: up-sampling -----------------------------------------------------------
    def up_sample(self, rays_o, rays_d, z_vals, sdf, r_limit, n_importance, inv_s):
        B, n = z_vals.shape
        return _C.neus_upsample(rays_o.contiguous(), rays_d.contiguous(), z_vals.contiguous(),
                                sdf.reshape(B, n).contiguous(), float(r_limit), float(inv_s),
                                self._quantiles(n_importance, z_vals.device))

    def cat_z_vals(self, rays_o, rays_d, z_vals, new_z_vals, sdf, last=False):
        if last:
            z, _ = _C.neus_merge(z_vals.contiguous(), None, new_z_vals.contiguous(), None)
            return z, sdf
        new_sdf = self._sdf_at(rays_o.contiguous(), rays_d.contiguous(), new_z_vals.contiguous()).reshape(new_z_vals.shape)
        return _C.neus_merge(z_vals.contiguous(), sdf.reshape(z_vals.shape).contiguous(), new_z_vals.contiguous(), new_sdf)

    def _sdf_pack(self):
        return self.sdf_network.packs(max_tiles=self.color_network.max_tiles(), mode=self.matrix_mode)

    def _sdf_at(self, rays_o, rays_d, z):
        """SDF at the ray samples: on the step's library-built x3 packs inside a training render that prepared them
        (NeusTrainEngine.prepare_step), else on the network's own cached packs."""
        if self._step_pack is not None:
            return _C.neus_sdf_points(None, None, rays_o=rays_o, rays_d=rays_d, z=z, mode='x3', pack=self._step_pack)
        wb_s, d_s = self._sdf_pack()
        return _C.neus_sdf_points(d_s, wb_s, rays_o=rays_o, rays_d=rays_d, z=z, mode=self.matrix_mode)

    @torch.no_grad()
    def _importance_z(self, rays_o, rays_d, z_vals, radius):
        B = rays_o.shape[0]
        sdf = self._sdf_at(rays_o, rays_d, z_vals).reshape(B, self.n_samples)
        m = self.n_importance // self.up_sample_steps
        for i in range(self.up_sample_steps):
            new_z = self.up_sample(rays_o, rays_d, z_vals, sdf, radius, m, 64 * 2 ** i)
            z_vals, sdf = self.cat_z_vals(rays_o, rays_d, z_vals, new_z, sdf, last=(i + 1 == self.up_sample_steps))
        return z_vals

    # ---- NeRF++ background (renderer.py:93-129).  Dead code in every shipped conf (n_outside = 0); kept for API
    # completeness as torch ops on the GPU -- not a HIP path (SURVEY 8a9').
    def render_core_outside(self, rays_o, rays_d, z_vals, sample_dist, nerf, background_rgb=None):
        B, n = z_vals.shape
        dists = z_vals[..., 1:] - z_vals[..., :-1]
        dists = torch.cat([dists, torch.full_like(dists[..., :1], float(sample_dist))], -1)
        mid = z_vals + dists * 0.5
        pts = rays_o[:, None, :] + rays_d[:, None, :] * mid[..., :, None]
        r = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).clip(1.0, 1e10)
        pts = torch.cat([pts / r, 1.0 / r], dim=-1).reshape(-1, 4)
        dirs = rays_d[:, None, :].expand(B, n, 3).reshape(-1, 3)
        density, col = nerf(pts, dirs)
        alpha = 1.0 - torch.exp(-F.softplus(density.reshape(B, n)) * dists)
        weights = alpha * torch.cumprod(torch.cat([torch.ones([B, 1], device=alpha.device), 1. - alpha + 1e-7], -1), -1)[:, :-1]
        col = col.reshape(B, n, 3)
        color = (weights[:, :, None] * col).sum(dim=1)
        if background_rgb is not None:
            color = color + background_rgb * (1.0 - weights.sum(dim=-1, keepdim=True))
        return {'color': color, 'sampled_color': col, 'alpha': alpha, 'weights': weights}

    # ---- render_core ----------------------------------------------------------------------------
    def render_core(self, rays_o, rays_d, z_vals, sample_dist, radius, sdf_network, deviation_network, color_network,
                    background_alpha=None, background_sampled_color=None, background_rgb=None,
                    cos_anneal_ratio=0.0, to_light=False):
        if background_alpha is not None:
            self.last_train_backend = 'torch'
            return self._render_core_autograd(rays_o, rays_d, z_vals, sample_dist, radius, sdf_network, deviation_network,
                                              color_network, background_rgb, cos_anneal_ratio, to_light,
                                              background_alpha=background_alpha, background_sampled_color=background_sampled_color)
        if _needs_graph(sdf_network, rays_o) or any(p.requires_grad and torch.is_grad_enabled()
                                                    for m in (deviation_network, color_network) for p in m.parameters()):
            if self.train_backend == 'hip' and rays_o.is_cuda and self._train_engine(sdf_network, color_network) is not None:
                self.last_train_backend = 'hip'
                return self._render_core_train_hip(rays_o, rays_d, z_vals, sample_dist, radius, sdf_network,
                                                   deviation_network, color_network, background_rgb, cos_anneal_ratio, to_light)
            self.last_train_backend = 'torch'
            return self._render_core_autograd(rays_o, rays_d, z_vals, sample_dist, radius, sdf_network,
                                              deviation_network, color_network, background_rgb, cos_anneal_ratio, to_light)
        B, n = z_vals.shape
        per_ray = sample_dist.reshape(-1).float().contiguous() if to_light else None
        mid_z, dists = _C.neus_section_mids(z_vals.contiguous(), 0.0 if to_light else float(sample_dist), per_ray)
        wb_s, d_s, wb_c, d_c = self._packs()
        if self.weights_only:
            # occupancy queries (gen_geo.py:231-242 uses nothing but weight_sum): the colour network is skipped
            no_col = np.zeros(packing.COL_DESC_INTS, np.int32)
            sdf, grad, _ = _C.neus_fine_points(d_s, wb_s, no_col, wb_s, rays_o=rays_o, rays_d=rays_d, z=mid_z, mode=self.matrix_mode)
            rgb = torch.zeros_like(grad)
        else:
            sdf, grad, rgb = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, rays_o=rays_o, rays_d=rays_d, z=mid_z, mode=self.matrix_mode)
        inv_s = torch.exp(deviation_network.variance.detach().float() * 10.0).reshape(1).contiguous()
        bg = None if background_rgb is None else background_rgb.detach().float().to(z_vals.device)
        o = _C.neus_composite_fwd(rays_o, rays_d, mid_z, dists, sdf, grad, rgb, inv_s, bg, radius, cos_anneal_ratio)
        gerr = o['gerr'].sum(0)
        inv_s_c = inv_s.clip(1e-6, 1e6)
        return {
            'color': o['color'], 'sdf': sdf.reshape(-1, 1), 'dists': dists, 'gradients': grad.reshape(B, n, 3),
            's_val': (1.0 / inv_s_c).reshape(1, 1).expand(B * n, 1), 'mid_z_vals': mid_z, 'weights': o['weights'],
            'cdf': o['cdf'], 'gradient_error': gerr[0] / (gerr[1] + 1e-5), 'inside_sphere': o['inside_sphere'],
            'surf': o['surf'], 'depth': o['depth'], 'weight_sum': o['weight_sum'], 'weight_max': o['weight_max'],
            'sampled_color': rgb.reshape(B, n, 3),
        }

    def _prepare_train_step(self, rays_o):
        """the step's x3 packs when this render is going to take the HIP training path with the exact-split forward (see
        NeusTrainEngine.prepare_step); None otherwise"""
        if self.n_outside > 0 or self.train_backend != 'hip' or not rays_o.is_cuda or not torch.is_grad_enabled():
            return None
        sn, cn, dn = self.sdf_network, self.color_network, self.deviation_network
        if not (_needs_graph(sn, rays_o) or any(p.requires_grad for m in (dn, cn) for p in m.parameters())):
            return None
        engine = self._train_engine(sn, cn)
        if engine is None:
            return None
        return engine.prepare_step([getattr(sn, 'lin%d' % l) for l in range(sn.num_layers - 1)],
                                   [getattr(cn, 'lin%d' % l) for l in range(cn.num_layers - 1)])

    # ---- training: explicit forward / backward tile programs (geo/train_programs.py) ----------------
    def _train_engine(self, sdf_network, color_network):
        key = (id(sdf_network), id(color_network))
        if key not in self._engines:
            try:
                from vqnerf_release_amd.geo.train_programs import NeusTrainEngine
                self._engines[key] = NeusTrainEngine(sdf_network, color_network)
            except AssertionError as e:
                # network shape outside what the tile programs cover -> torch autograd path, and say so
                warnings.warn('NeuSRenderer: the tile-program training engine does not cover this network shape '
                              f'({e or "shape assertion in NeusTrainEngine"}); training runs on the torch-autograd statement of '
                              'render_core instead (renderer.last_train_backend == "torch")', RuntimeWarning, stacklevel=3)
                self._engines[key] = None
        return self._engines[key]

    def _render_core_train_hip(self, rays_o, rays_d, z_vals, sample_dist, radius, sdf_network, deviation_network,
                               color_network, background_rgb, cos_anneal_ratio, to_light):
        from vqnerf_release_amd.geo.train_programs import NeusCoreFunction
        B, n = z_vals.shape
        per_ray = sample_dist.reshape(-1).float().contiguous() if to_light else None
        mid_z, dists = _C.neus_section_mids(z_vals.detach().contiguous(), 0.0 if to_light else float(sample_dist), per_ray)
        pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]).reshape(-1, 3)
        dirs = rays_d[:, None, :].expand(B, n, 3).reshape(-1, 3)
        engine = self._train_engine(sdf_network, color_network)
        s_lins = [getattr(sdf_network, 'lin%d' % l) for l in range(sdf_network.num_layers - 1)]
        c_lins = [getattr(color_network, 'lin%d' % l) for l in range(color_network.num_layers - 1)]
        from vqnerf_release_amd.geo.models.fields import effective_weights
        ws = effective_weights(list(s_lins) + list(c_lins))          # all weight-norm chains of the step: one launch each way
        params = ws[:len(s_lins)] + [m.bias for m in s_lins] + ws[len(s_lins):] + [m.bias for m in c_lins]
        sdf, grad, rgb = NeusCoreFunction.apply(engine, pts, dirs, *params)
        inv_s = torch.exp(deviation_network.variance * 10.0).reshape(1)
        bg = None if background_rgb is None else background_rgb.detach().float().to(z_vals.device)
        color, wsum, gerr, weights, cdf, inside, surf, depth, wmax = CompositeFunction.apply(
            sdf.reshape(B, n), grad.reshape(B, n, 3), rgb.reshape(B, n, 3), inv_s, rays_o, rays_d, mid_z, dists, bg,
            float(radius), float(cos_anneal_ratio))
        return {
            'color': color, 'sdf': sdf.reshape(-1, 1), 'dists': dists, 'gradients': grad.reshape(B, n, 3),
            's_val': (1.0 / inv_s.clip(1e-6, 1e6)).reshape(1, 1).expand(B * n, 1), 'mid_z_vals': mid_z, 'weights': weights,
            'cdf': cdf, 'gradient_error': gerr, 'inside_sphere': inside, 'surf': surf, 'depth': depth, 'weight_sum': wsum,
            'weight_max': wmax, 'sampled_color': rgb.reshape(B, n, 3),
        }

    def _render_core_autograd(self, rays_o, rays_d, z_vals, sample_dist, radius, sdf_network, deviation_network,
                              color_network, background_rgb, cos_anneal_ratio, to_light, background_alpha=None,
                              background_sampled_color=None):
        """torch-op statement of renderer.py:207-297 for when autograd needs the graph."""
        B, n = z_vals.shape
        dists = z_vals[..., 1:] - z_vals[..., :-1]
        tail = sample_dist if to_light else torch.full_like(dists[..., :1], float(sample_dist))
        dists = torch.cat([dists, tail], -1)
        mid_z = z_vals + dists * 0.5
        pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]).reshape(-1, 3)
        dirs = rays_d[:, None, :].expand(B, n, 3).reshape(-1, 3)
        out = sdf_network(pts)
        sdf, feat = out[:, :1], out[:, 1:]
        grads = sdf_network.gradient(pts).squeeze(1)
        rgb = color_network(pts, grads, dirs, feat).reshape(B, n, 3)
        inv_s = deviation_network(torch.zeros([1, 3], device=pts.device))[:, :1].clip(1e-6, 1e6)
        true_cos = (dirs * grads).sum(-1, keepdim=True)
        iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + F.relu(-true_cos) * cos_anneal_ratio)
        d = dists.reshape(-1, 1)
        prev_cdf = torch.sigmoid((sdf - iter_cos * d * 0.5) * inv_s)
        next_cdf = torch.sigmoid((sdf + iter_cos * d * 0.5) * inv_s)
        alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).reshape(B, n).clip(0.0, 1.0)
        pts_r = torch.linalg.norm(pts, ord=2, dim=-1).reshape(B, n)
        inside = (pts_r < radius).float().detach()
        relax = (pts_r < radius * 1.1).float().detach()
        if background_alpha is not None:                      # renderer.py:254-261
            alpha = alpha * inside + background_alpha[:, :n] * (1.0 - inside)
            alpha = torch.cat([alpha, background_alpha[:, n:]], dim=-1)
            rgb = rgb * inside[:, :, None] + background_sampled_color[:, :n] * (1.0 - inside)[:, :, None]
            rgb = torch.cat([rgb, background_sampled_color[:, n:]], dim=1)
        trans = torch.cumprod(torch.cat([torch.ones([B, 1], device=alpha.device), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
        weights = alpha * trans
        wsum = weights.sum(-1, keepdim=True)
        color = (rgb * weights[:, :, None]).sum(1)
        surf = (pts.reshape(B, n, 3) * weights[:, :n, None]).sum(1)
        depth = torch.linalg.norm(surf - rays_o, ord=2, dim=-1, keepdim=True)
        if background_rgb is not None:
            color = color + background_rgb * (1.0 - wsum)
        g3 = grads.reshape(B, n, 3)
        gerr = (torch.linalg.norm(g3, ord=2, dim=-1) - 1.0) ** 2
        gerr = (relax * gerr).sum() / (relax.sum() + 1e-5)
        return {
            'color': color, 'sdf': sdf, 'dists': dists, 'gradients': g3, 's_val': (1.0 / inv_s).expand(B * n, 1),
            'mid_z_vals': mid_z, 'weights': weights, 'cdf': prev_cdf.reshape(B, n), 'gradient_error': gerr,
            'inside_sphere': inside, 'surf': surf, 'depth': depth, 'weight_sum': wsum,
            'weight_max': weights.max(-1, keepdim=True)[0], 'sampled_color': rgb,
        }

    # ---- render ---------------------------------------------------------------------------------
    def render(self, rays_o, rays_d, near, far, radius, perturb_overwrite=-1, background_rgb=None,
               cos_anneal_ratio=0.0, to_light=False, t_rand=None):
        rays_o = rays_o.float().contiguous()
        rays_d = rays_d.float().contiguous()
        dev = rays_o.device
        B = len(rays_o)
        sample_dist = (far - near) / self.n_samples if to_light else 2 * radius / self.n_samples
        z = torch.linspace(0.0, 1.0, self.n_samples, device=dev)
        z_vals = near + (far - near) * z[None, :]
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        if perturb > 0:
            if t_rand is None:
                t_rand = torch.rand([B, 1], device=dev) - 0.5
            z_vals = z_vals + t_rand * 2.0 * radius / self.n_samples
        z_vals = z_vals.float().contiguous()
        z_vals_outside = None
        if self.n_outside > 0:                                   # renderer.py:309-331 (torch ops; never reached by shipped confs)
            zo = torch.linspace(1e-3, 1.0 - 1.0 / (self.n_outside + 1.0), self.n_outside, device=dev)
            if perturb > 0:
                mids = .5 * (zo[..., 1:] + zo[..., :-1])
                upper, lower = torch.cat([mids, zo[..., -1:]], -1), torch.cat([zo[..., :1], mids], -1)
                zo = lower[None, :] + (upper - lower)[None, :] * torch.rand([B, zo.shape[-1]], device=dev)
            z_vals_outside = far / torch.flip(zo, dims=[-1]) + 1.0 / self.n_samples
        n = self.n_samples
        if self.n_importance > 0:
            self._step_pack = self._prepare_train_step(rays_o)
            try:
                z_vals = self._importance_z(rays_o, rays_d, z_vals, radius)
            finally:
                self._step_pack = None
            n = self.n_samples + self.n_importance
        bg_alpha = bg_color = None
        if self.n_outside > 0:
            z_feed, _ = torch.sort(torch.cat([z_vals, z_vals_outside.expand(B, -1)], dim=-1), dim=-1)
            ro = self.render_core_outside(rays_o, rays_d, z_feed, sample_dist, self.nerf)
            bg_color, bg_alpha = ro['sampled_color'], ro['alpha']
        rc = self.render_core(rays_o, rays_d, z_vals, sample_dist, radius, self.sdf_network, self.deviation_network,
                              self.color_network, background_alpha=bg_alpha, background_sampled_color=bg_color,
                              background_rgb=background_rgb, cos_anneal_ratio=cos_anneal_ratio, to_light=to_light)
        weights = rc['weights']
        return {
            'color_fine': rc['color'],
            's_val': rc['s_val'].reshape(B, -1).mean(dim=-1, keepdim=True),
            'cdf_fine': rc['cdf'],
            'weight_sum': rc['weight_sum'],
            'weight_max': rc['weight_max'],
            'gradients': rc['gradients'],
            'weights': weights,
            'gradient_error': rc['gradient_error'],
            'inside_sphere': rc['inside_sphere'],
            'surf': rc['surf'],
            'depth': rc['depth'],
        }

    def extract_geometry(self, bound_min, bound_max, resolution, threshold=0.0):
        return extract_geometry(bound_min, bound_max, resolution=resolution, threshold=threshold,
                                query_func=lambda pts: -self.sdf_network.sdf(pts))
