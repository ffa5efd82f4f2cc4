"""Torch-CPU oracle for the NeuS ray marcher (geo half).  TEST INFRASTRUCTURE.

A functional restatement (weights are plain dicts of arrays, no nn.Module) of
    geo/NeuS-ours2/models/embedder.py:6-51     -> posenc
    geo/NeuS-ours2/models/fields.py:9-107      -> sdf_forward / sdf_gradient
    geo/NeuS-ours2/models/fields.py:111-172    -> color_forward
    geo/NeuS-ours2/models/fields.py:257-263    -> inv_s_from_variance
    geo/NeuS-ours2/models/renderer.py:39-69    -> sample_pdf_det
    geo/NeuS-ours2/models/renderer.py:131-175  -> up_sample
    geo/NeuS-ours2/models/renderer.py:177-191  -> cat_z_vals
    geo/NeuS-ours2/models/renderer.py:193-297  -> render_core
    geo/NeuS-ours2/models/renderer.py:299-401  -> render
    geo/NeuS-ours2/gen_geo.py:182-257          -> compute_vis        (restated from the text: gen_geo.py is not importable
    geo/NeuS-ours2/gen_geo.py:259-344          -> compute_geo         here -- cv2 / pyhocon / trimesh absent -- so these two
    geo/NeuS-ours2/gen_geo.py:346-369          -> intersect_circle, normal_correct_np, _np_norm    are unpinned callers of the pinned `render`)
It is *faithful*: it keeps the reference's redundant second SDF forward inside
``sdf_gradient`` (fields.py:98) and its op order, because it doubles as the
timed CPU baseline.  Pinned against the real reference by
``oracle/gen_golden_geo.py`` -> ``tests/golden/geo_*.npz``.

All functions take/return torch CPU tensors; dtype follows the inputs, so the
same code evaluated in float64 serves as "ground truth" for tolerance studies.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# configs (mirror geo/confs/*.conf `model{}` blocks)
# ----------------------------------------------------------------------------

FULL_CFG = dict(
    sdf=dict(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6,
             bias=0.5, scale=1.0),
    color=dict(d_feature=256, mode='idr', d_in=9, d_out=3, d_hidden=256, n_layers=4,
               multires_view=4, squeeze_out=True),
    variance=dict(init_val=0.3),
    renderer=dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0),
)

# BASELINE.json configs[0]: "2-layer-64 MLP", same wiring as the full nets
SMALL_CFG = dict(
    sdf=dict(d_in=3, d_out=65, d_hidden=64, n_layers=2, skip_in=(), multires=6,
             bias=0.5, scale=1.0),
    color=dict(d_feature=64, mode='idr', d_in=9, d_out=3, d_hidden=64, n_layers=2,
               multires_view=4, squeeze_out=True),
    variance=dict(init_val=0.3),
    renderer=dict(n_samples=64, n_importance=0, n_outside=0, up_sample_steps=4, perturb=1.0),
)


def sdf_dims(cfg):
    c = cfg['sdf']
    d0 = c['d_in'] + (c['d_in'] * 2 * c['multires'] if c['multires'] > 0 else 0)
    return [d0] + [c['d_hidden']] * c['n_layers'] + [c['d_out']]


def color_dims(cfg):
    c = cfg['color']
    d0 = c['d_in'] + c['d_feature']
    if c['multires_view'] > 0:
        d0 += 3 * 2 * c['multires_view']
    return [d0] + [c['d_hidden']] * c['n_layers'] + [c['d_out']]


# ----------------------------------------------------------------------------
# seeded weights (numpy RNG, so the reference side can load the very same arrays)
# ----------------------------------------------------------------------------

def make_sdf_params(cfg, seed=0):
    """weight_g / weight_v / bias per layer, drawn like fields.py:45-63
    (geometric init => a rough sphere of radius `bias`) but from numpy's RNG."""
    rng = np.random.default_rng(seed)
    c = cfg['sdf']
    dims = sdf_dims(cfg)
    n_lin = len(dims) - 1
    p = {}
    for l in range(n_lin):
        out_dim = dims[l + 1] - dims[0] if (l + 1) in c['skip_in'] else dims[l + 1]
        in_dim = dims[l]
        if l == n_lin - 1:
            w = rng.normal(math.sqrt(math.pi) / math.sqrt(in_dim), 1e-4, (out_dim, in_dim))
            b = np.full((out_dim,), -c['bias'])
        elif c['multires'] > 0 and l == 0:
            w = np.zeros((out_dim, in_dim))
            w[:, :3] = rng.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), (out_dim, 3))
            b = np.zeros((out_dim,))
        elif c['multires'] > 0 and l in c['skip_in']:
            w = rng.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), (out_dim, in_dim))
            w[:, -(dims[0] - 3):] = 0.0
            b = np.zeros((out_dim,))
        else:
            w = rng.normal(0.0, math.sqrt(2) / math.sqrt(out_dim), (out_dim, in_dim))
            b = np.zeros((out_dim,))
        # perturb so that every weight matters in parity tests (trained-net-like)
        w = w + rng.normal(0.0, 0.02 / math.sqrt(in_dim), w.shape)
        b = b + rng.normal(0.0, 0.01, b.shape)
        v = w.astype(np.float32)
        g = np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True)
        g = (g * rng.uniform(0.9, 1.1, g.shape)).astype(np.float32)
        p[f'lin{l}.weight_g'] = g
        p[f'lin{l}.weight_v'] = v
        p[f'lin{l}.bias'] = b.astype(np.float32)
    return p


def make_color_params(cfg, seed=1):
    rng = np.random.default_rng(seed)
    dims = color_dims(cfg)
    p = {}
    for l in range(len(dims) - 1):
        k = 1.0 / math.sqrt(dims[l])
        v = rng.uniform(-k, k, (dims[l + 1], dims[l])).astype(np.float32)
        g = np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True)
        g = (g * rng.uniform(0.8, 1.6, g.shape)).astype(np.float32)
        p[f'lin{l}.weight_g'] = g
        p[f'lin{l}.weight_v'] = v
        p[f'lin{l}.bias'] = rng.uniform(-k, k, (dims[l + 1],)).astype(np.float32)
    return p


def make_rays(n_rays, seed=2, cam=(0.0, 0.0, 4.0), spread=0.35):
    """Pin-hole-ish rays from `cam` towards the origin; ~half hit a 0.5-sphere."""
    rng = np.random.default_rng(seed)
    o = np.tile(np.asarray(cam, np.float32)[None], (n_rays, 1))
    tgt = rng.uniform(-1.0, 1.0, (n_rays, 3)) * spread * 4.0 * np.array([1, 1, 0])
    d = tgt - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    near = np.full((n_rays, 1), 2.0, np.float32)
    far = np.full((n_rays, 1), 6.0, np.float32)
    return o.astype(np.float32), d.astype(np.float32), near, far


def make_hit_rays(n_hit=56, n_miss=8, seed=21):
    """Inputs of tests/golden/geo_hits.npz: `n_hit` rays aimed at the disc the 0.5-sphere projects to (about half of them hit
    it; with variance 0.5, i.e. inv_s = e^5, those reach weight_sum ~ 1) followed by `n_miss` rays that leave the radius-2
    bounding sphere untouched (all `inside_sphere` flags 0: the all-masked branch of up_sample, renderer.py:137-164).
    Also returns the per-ray near / far of the to_light case and the injected jitter (renderer.py:318, `torch.rand([B,1]) - 0.5`)."""
    o, d, near, far = make_rays(n_hit, seed=seed, spread=0.12)
    rng = np.random.default_rng(seed + 1)
    th = np.deg2rad(rng.uniform(40.0, 70.0, n_miss))
    ph = rng.uniform(0.0, 2 * np.pi, n_miss)
    dm = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), -np.cos(th)], -1).astype(np.float32)
    o = np.concatenate([o, np.tile(o[:1], (n_miss, 1))], 0)
    d = np.concatenate([d, dm], 0)
    B = n_hit + n_miss
    near = np.full((B, 1), 2.0, np.float32)
    far = np.full((B, 1), 6.0, np.float32)
    near_l = rng.uniform(1.5, 2.5, (B, 1)).astype(np.float32)
    far_l = (near_l + rng.uniform(3.0, 4.0, (B, 1))).astype(np.float32)
    t_rand = (rng.uniform(0.0, 1.0, (B, 1)) - 0.5).astype(np.float32)
    return dict(o=o, d=d, near=near, far=far, near_l=near_l, far_l=far_l, t_rand=t_rand)


def make_upsample_edge_inputs(n, seed=30):
    """Inputs of tests/golden/geo_upsample_edge.npz: 8 rays x n sorted depths with hand-made SDF profiles that drive
    `up_sample` + `sample_pdf` (renderer.py:131-175, :39-69) into their edge branches:
      0 ray that misses the bounding sphere (every `inside_sphere` 0 -> cos clipped to 0 -> near-uniform weights)
      1 sdf = +5 everywhere (both sigmoids saturate at 1: alpha = 1e-5 / (1 + 1e-5) in every section)
      2 sdf = -5 everywhere (both sigmoids 0: alpha = 1 in section 0 -> one spike, all later cdf steps < 1e-5 -> `denom` = 1)
      3 one steep zero crossing between samples 5 and 6 (single spike in the interior)
      4 gentle linear decrease through zero (the regular case)
      5 zero crossing in the very last section
      6 in - out - in (two crossings)
      7 smooth random profile."""
    rng = np.random.default_rng(seed + n)
    o = np.tile(np.array([[0.0, 0.0, 4.0]], np.float32), (8, 1))
    d = np.tile(np.array([[0.0, 0.0, -1.0]], np.float32), (8, 1))
    d[0] = np.array([np.sin(1.0), 0.0, -np.cos(1.0)], np.float32)
    z = np.sort(np.concatenate([np.tile(np.linspace(2.0, 6.0, 64, dtype=np.float32)[None], (8, 1)),
                                rng.uniform(2.0, 6.0, (8, n - 64)).astype(np.float32)], 1), 1)
    t = (z - 2.0) / 4.0
    sdf = np.zeros((8, n), np.float32)
    sdf[0] = rng.normal(0.0, 0.3, n)
    sdf[1] = 5.0
    sdf[2] = -5.0
    sdf[3] = np.where(np.arange(n) <= 5, 0.5, -0.5)
    sdf[4] = 0.4 - 0.8 * t[4]
    sdf[5] = np.where(np.arange(n) < n - 1, 0.3, -0.3)
    sdf[6] = 0.25 * np.cos(4 * np.pi * t[6])
    sdf[7] = np.convolve(rng.normal(0.0, 0.2, n + 8), np.ones(9) / 9.0, 'valid')
    return o, d, z.astype(np.float32), sdf.astype(np.float32)


def to_torch(params, dtype=torch.float32):
    return {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in params.items()}


# ----------------------------------------------------------------------------
# networks
# ----------------------------------------------------------------------------

def posenc(x, n_freqs):
    """[x, sin(2^0 x), cos(2^0 x), ..., sin(2^(n-1) x), cos(2^(n-1) x)]  (embedder.py:16-34)."""
    if n_freqs <= 0:
        return x
    outs = [x]
    freqs = 2.0 ** torch.linspace(0.0, n_freqs - 1, n_freqs)
    for f in freqs:
        f = f.to(x.dtype)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


def wn_weight(p, l):
    """nn.utils.weight_norm effective weight: g * v / ||v||_row  (fields.py:65-66)."""
    g, v = p[f'lin{l}.weight_g'], p[f'lin{l}.weight_v']
    return g * v / v.norm(dim=1, keepdim=True)


def softplus100(x):
    return F.softplus(x, beta=100)  # default threshold=20, as nn.Softplus(beta=100)


def sdf_forward(p, cfg, pts):
    """fields.py:72-91.  returns [P, d_out] = [sdf/scale, feature]."""
    c = cfg['sdf']
    n_lin = len(sdf_dims(cfg)) - 1
    inputs = pts * c['scale']
    if c['multires'] > 0:
        inputs = posenc(inputs, c['multires'])
    x = inputs
    for l in range(n_lin):
        if l in c['skip_in']:
            x = torch.cat([x, inputs], 1) / math.sqrt(2)
        x = F.linear(x, wn_weight(p, l), p[f'lin{l}.bias'])
        if l < n_lin - 1:
            x = softplus100(x)
    return torch.cat([x[:, :1] / c['scale'], x[:, 1:]], -1)


def sdf_only(p, cfg, pts):
    return sdf_forward(p, cfg, pts)[:, :1]


def sdf_gradient(p, cfg, pts, create_graph=False):
    """fields.py:96-107: a *second* full forward, then autograd wrt the input."""
    x = pts.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        y = sdf_only(p, cfg, x)
        g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=create_graph,
                                retain_graph=create_graph)[0]
    return g if create_graph else g.detach()


def color_forward(p, cfg, pts, normals, dirs, feats):
    """fields.py:147-172 (mode 'idr' and the two ablation modes)."""
    c = cfg['color']
    n_lin = len(color_dims(cfg)) - 1
    if c['multires_view'] > 0:
        dirs = posenc(dirs, c['multires_view'])
    if c['mode'] == 'idr':
        x = torch.cat([pts, dirs, normals, feats], -1)
    elif c['mode'] == 'no_view_dir':
        x = torch.cat([pts, normals, feats], -1)
    else:
        x = torch.cat([pts, dirs, feats], -1)
    for l in range(n_lin):
        x = F.linear(x, wn_weight(p, l), p[f'lin{l}.bias'])
        if l < n_lin - 1:
            x = F.relu(x)
    return torch.sigmoid(x) if c['squeeze_out'] else x


def inv_s_from_variance(variance):
    """fields.py:262-263 + the clip at renderer.py:229."""
    v = torch.as_tensor(variance)
    return torch.exp(v * 10.0).clip(1e-6, 1e6)


# ----------------------------------------------------------------------------
# per-ray sampling / compositing
# ----------------------------------------------------------------------------

def sample_pdf_det(bins, weights, n_new):
    """renderer.py:39-69 with det=True."""
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = torch.linspace(0.5 / n_new, 1.0 - 0.5 / n_new, n_new, dtype=torch.float32).to(cdf.dtype)
    u = u.expand(list(cdf.shape[:-1]) + [n_new]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_b, bin_a = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


def up_sample_weights(rays_o, rays_d, z_vals, sdf, r_limit, inv_s):
    """renderer.py:135-172: the importance weights the new samples are drawn from."""
    B, n = z_vals.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., None]
    rad = torch.linalg.norm(pts, ord=2, dim=-1)
    inside = (rad[:, :-1] < r_limit) | (rad[:, 1:] < r_limit)
    sdf = sdf.reshape(B, n)
    ps, ns = sdf[:, :-1], sdf[:, 1:]
    pz, nz = z_vals[:, :-1], z_vals[:, 1:]
    mid = (ps + ns) * 0.5
    cos = (ns - ps) / (nz - pz + 1e-5)
    prev_cos = torch.cat([torch.zeros(B, 1, dtype=cos.dtype), cos[:, :-1]], -1)
    cos = torch.minimum(prev_cos, cos)
    cos = cos.clip(-1e3, 0.0) * inside
    dist = nz - pz
    p_est = mid - cos * dist * 0.5
    n_est = mid + cos * dist * 0.5
    p_cdf = torch.sigmoid(p_est * inv_s)
    n_cdf = torch.sigmoid(n_est * inv_s)
    alpha = (p_cdf - n_cdf + 1e-5) / (p_cdf + 1e-5)
    trans = torch.cumprod(torch.cat([torch.ones(B, 1, dtype=alpha.dtype), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    return alpha * trans


def up_sample(rays_o, rays_d, z_vals, sdf, r_limit, n_new, inv_s):
    w = up_sample_weights(rays_o, rays_d, z_vals, sdf, r_limit, inv_s)
    return sample_pdf_det(z_vals, w, n_new)


def cat_z_vals(p_sdf, cfg, rays_o, rays_d, z_vals, new_z, sdf, last):
    """renderer.py:177-191.  Returns (z_sorted, sdf_sorted, had_ties).  `had_ties` reports equal depths only where they
    matter: the order `torch.sort` gives equal keys is unspecified and decides how `sdf` is permuted (:187-189); in the last
    step nothing is permuted and the sorted depths are the same values either way."""
    B, n = z_vals.shape
    m = new_z.shape[1]
    z_cat = torch.cat([z_vals, new_z], -1)
    z_sorted, index = torch.sort(z_cat, dim=-1)
    ties = (not last) and bool((z_sorted[:, 1:] == z_sorted[:, :-1]).any())
    if not last:
        pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z[..., None]
        new_sdf = sdf_only(p_sdf, cfg, pts.reshape(-1, 3)).reshape(B, m)
        sdf = torch.gather(torch.cat([sdf, new_sdf], -1), -1, index)
    return z_sorted, sdf, ties


def render_core(p_sdf, p_col, variance, cfg, rays_o, rays_d, z_vals, sample_dist, radius,
                background_rgb=None, cos_anneal_ratio=0.0, create_graph=False, to_light=False):
    """renderer.py:193-297 (n_outside == 0 branch).  to_light: `sample_dist` is the per-ray [B,1] tensor of :302 (:211)."""
    B, n = z_vals.shape
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    tail = sample_dist if to_light else torch.full_like(dists[..., :1], float(sample_dist))
    dists = torch.cat([dists, tail], -1)
    mid_z = z_vals + dists * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(B, n, 3).reshape(-1, 3)

    out = sdf_forward(p_sdf, cfg, pts)
    sdf, feat = out[:, :1], out[:, 1:]
    grads = sdf_gradient(p_sdf, cfg, pts, create_graph=create_graph)
    rgb = color_forward(p_col, cfg, pts, grads, dirs, feat).reshape(B, n, 3)

    inv_s = inv_s_from_variance(variance).to(sdf.dtype).reshape(1, 1)
    true_cos = (dirs * grads).sum(-1, keepdim=True)
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio)
                 + F.relu(-true_cos) * cos_anneal_ratio)
    d = dists.reshape(-1, 1)
    est_next = sdf + iter_cos * d * 0.5
    est_prev = sdf - iter_cos * d * 0.5
    prev_cdf = torch.sigmoid(est_prev * inv_s)
    next_cdf = torch.sigmoid(est_next * inv_s)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).reshape(B, n).clip(0.0, 1.0)

    pts_r = torch.linalg.norm(pts, ord=2, dim=-1).reshape(B, n)
    inside = (pts_r < radius).to(sdf.dtype)
    relax = (pts_r < radius * 1.1).to(sdf.dtype)

    trans = torch.cumprod(torch.cat([torch.ones(B, 1, dtype=alpha.dtype), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    weights = alpha * trans
    wsum = weights.sum(-1, keepdim=True)
    color = (rgb * weights[..., None]).sum(1)
    surf = (pts.reshape(B, n, 3) * weights[..., None]).sum(1)
    depth = torch.linalg.norm(surf - rays_o, ord=2, dim=-1, keepdim=True)
    if background_rgb is not None:
        color = color + background_rgb * (1.0 - wsum)
    g3 = grads.reshape(B, n, 3)
    gerr = (torch.linalg.norm(g3, ord=2, dim=-1) - 1.0) ** 2
    gerr = (relax * gerr).sum() / (relax.sum() + 1e-5)
    return dict(color=color, sdf=sdf, dists=dists, gradients=g3,
                s_val=(1.0 / inv_s).expand(B * n, 1), mid_z_vals=mid_z, weights=weights,
                cdf=prev_cdf.reshape(B, n), gradient_error=gerr, inside_sphere=inside,
                surf=surf, depth=depth, sampled_color=rgb, alpha=alpha)


def coarse_to_fine_z(p_sdf, cfg, rays_o, rays_d, z_vals, radius):
    """The no_grad up-sampling loop, renderer.py:335-353.  Returns (z_fine, had_ties)."""
    r = cfg['renderer']
    B = rays_o.shape[0]
    ties = False
    with torch.no_grad():
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., None]
        sdf = sdf_only(p_sdf, cfg, pts.reshape(-1, 3)).reshape(B, r['n_samples'])
        steps = r['up_sample_steps']
        for i in range(steps):
            new_z = up_sample(rays_o, rays_d, z_vals, sdf, radius, r['n_importance'] // steps, 64 * 2 ** i)
            z_vals, sdf, t = cat_z_vals(p_sdf, cfg, rays_o, rays_d, z_vals, new_z, sdf, last=(i + 1 == steps))
            ties = ties or t
    return z_vals, ties


def render(p_sdf, p_col, variance, cfg, rays_o, rays_d, near, far, radius, jitter=None,
           background_rgb=None, cos_anneal_ratio=0.0, create_graph=False, to_light=False):
    """renderer.py:299-401 (n_outside == 0).  to_light: per-ray last-section length (far - near) / n_samples (:302).

    `jitter` is the explicit stand-in for ``torch.rand([B,1]) - 0.5`` at
    renderer.py:318 (None == perturb 0)."""
    r = cfg['renderer']
    B = rays_o.shape[0]
    sample_dist = (far - near) / r['n_samples'] if to_light else 2.0 * radius / r['n_samples']
    z = torch.linspace(0.0, 1.0, r['n_samples'], dtype=torch.float32).to(rays_o.dtype)
    z_vals = near + (far - near) * z[None, :]
    if jitter is not None:
        z_vals = z_vals + jitter * 2.0 * radius / r['n_samples']
    ties = False
    if r['n_importance'] > 0:
        z_vals, ties = coarse_to_fine_z(p_sdf, cfg, rays_o, rays_d, z_vals, radius)
    n = r['n_samples'] + r['n_importance']
    rc = render_core(p_sdf, p_col, variance, cfg, rays_o, rays_d, z_vals, sample_dist, radius,
                     background_rgb=background_rgb, cos_anneal_ratio=cos_anneal_ratio,
                     create_graph=create_graph, to_light=to_light)
    w = rc['weights']
    return dict(color_fine=rc['color'],
                s_val=rc['s_val'].reshape(B, n).mean(-1, keepdim=True),
                cdf_fine=rc['cdf'], weight_sum=w.sum(-1, keepdim=True),
                weight_max=w.max(-1, keepdim=True)[0], gradients=rc['gradients'],
                weights=w, gradient_error=rc['gradient_error'],
                inside_sphere=rc['inside_sphere'], surf=rc['surf'], depth=rc['depth'],
                z_vals=z_vals, had_ties=ties)


# ----------------------------------------------------------------------------
# per-view geometry / light-visibility extraction (geo/NeuS-ours2/gen_geo.py) -- SURVEY 8(f1)
# gen_geo.py itself cannot be imported here (cv2, pyhocon, trimesh are absent), so this is a restatement from its text;
# it runs on the pinned `render` above.  The reference renders with the conf's `perturb = 1` (a fresh torch.rand jitter per
# call, gen_geo.py:229-235 / :275-281 pass no perturb_overwrite): here the jitter is an explicit argument (None = none).
# ----------------------------------------------------------------------------

def intersect_circle(x, d, r, eps=1e-7):
    """gen_geo.py:346-357: the larger root of |x + t d| = r, and the point there."""
    b = 2.0 * torch.sum(x * d, dim=-1)
    a = torch.sum(d * d, dim=-1)
    c = torch.sum(x * x, dim=-1) - r ** 2
    eps = torch.ones_like(a) * eps
    denom = torch.where(2 * a > eps, 2 * a, eps)
    t1 = (-b + torch.sqrt(torch.square(b) - 4.0 * a * c)) / denom
    t2 = (-b - torch.sqrt(torch.square(b) - 4.0 * a * c)) / denom
    t = torch.where(t1 > t2, t1, t2)
    return t[:, None], x + t[:, None] * d


def _np_norm(src, dim):
    """gen_geo.py:367-369."""
    return src / np.sqrt(np.sum(np.square(src), axis=dim, keepdims=True))


def normal_correct_np(rays_o, surf, normal):
    """gen_geo.py:358-365: flip normals that face away from the camera."""
    surf2c = rays_o - surf
    surf2c = surf2c / np.linalg.norm(surf2c, ord=2, axis=-1, keepdims=True)
    cos = np.sum(surf2c * normal, axis=-1, keepdims=True)
    return np.where(cos >= 0.0, normal, -normal)


def compute_geo(p_sdf, p_col, variance, cfg, rays_o, rays_d, near, far, max_radius, alpha_thres=0.5, use_white_bkgd=True,
                cos_anneal_ratio=1.0, batch_size=512, jitter=None):
    """gen_geo.py:259-344 for the rays of one view, flattened [R,3] (the file / image writing at :329-342 is not restated).
    Returns numpy arrays: rgb [R,3] (`color_fine`), surf [R,3], normal [R,3] (`rot_normal`: the weight- and inside-sphere-
    weighted mean gradient, normalised, camera-facing, the unit diagonal on background pixels :321-326), mask [R,1]."""
    out_rgb, out_normal, out_surf, out_mask = [], [], [], []
    R = rays_o.shape[0]
    for s in range(0, R, batch_size):
        o, d = rays_o[s:s + batch_size], rays_d[s:s + batch_size]
        bg = torch.ones(1, 3) if use_white_bkgd else None
        ro = render(p_sdf, p_col, variance, cfg, o, d, near[s:s + batch_size], far[s:s + batch_size], max_radius,
                    jitter=None if jitter is None else jitter[s:s + batch_size], background_rgb=bg, cos_anneal_ratio=cos_anneal_ratio)
        out_rgb.append(ro['color_fine'].detach().numpy())
        alpha_mask = ro['weight_sum'].detach().numpy()
        out_mask.append(np.where(alpha_mask > alpha_thres, 1.0, 0.0))
        surf = ro['surf'].detach().numpy()
        out_surf.append(surf)
        n_samples = cfg['renderer']['n_samples'] + cfg['renderer']['n_importance']
        normals = ro['gradients'] * ro['weights'][:, :n_samples, None]
        normals = normals * ro['inside_sphere'][..., None]
        normals = _np_norm(normals.sum(dim=1).detach().numpy(), dim=-1)
        out_normal.append(normal_correct_np(o.numpy(), surf, normals))
    rgb, surf = np.concatenate(out_rgb, 0), np.concatenate(out_surf, 0)
    img_mask = (np.concatenate(out_mask, 0) * 256).clip(0, 255)                       # :318-319 (the 8-bit mask image)
    normal = np.concatenate(out_normal, 0)
    rot_normal = normal * (img_mask / 255.0) + _np_norm(np.ones_like(normal), dim=-1) * (1.0 - img_mask / 255.0)
    return dict(rgb=rgb, surf=surf, normal=rot_normal, mask=img_mask / 255.0)


def compute_vis(p_sdf, p_col, variance, cfg, lxyz_flat, surf, normal, mask, max_radius, use_white_bkgd=True, cos_anneal_ratio=1.0,
                batch_size=512, lpix_chunk=1):
    """gen_geo.py:182-257 for one view, flattened: surf / normal [R,3], mask [R,1], lxyz_flat [1,L,3] -> lvis [R,L] float32
    (1 - weight_sum of the secondary ray towards every FRONT-LIT light of every foreground pixel, zeros elsewhere).  Walks the
    lights `lpix_chunk` at a time like the reference (one `render` per light by default)."""
    n_lights = lxyz_flat.shape[1]
    alpha = mask[..., 0] > 0.0
    surf_fg, normal_fg = surf[alpha], normal[alpha]
    batch_lvis = []
    for surf_batch, normal_batch in zip(surf_fg.split(batch_size), normal_fg.split(batch_size)):
        lvis_hit = np.zeros((surf_batch.shape[0], n_lights), dtype=np.float32)
        for i in range(0, n_lights, lpix_chunk):
            end_i = min(n_lights, i + lpix_chunk)
            lxyz_chunk = lxyz_flat[:, i:end_i, :]
            surf2l = lxyz_chunk - surf_batch[:, None, :]
            surf2l = surf2l / torch.linalg.norm(surf2l, ord=2, dim=-1, keepdim=True)
            surf2l_flat = surf2l.reshape((-1, 3))
            surf_flat = surf_batch[:, None, :].repeat(1, surf2l.shape[1], 1).reshape((-1, 3))
            lcos = torch.einsum('ijk,ik->ij', surf2l, normal_batch)
            front_lit = lcos > 0
            if torch.sum(front_lit) == 0:
                continue
            front_lit_flat = front_lit.reshape((-1,))
            o, d = surf_flat[front_lit_flat], surf2l_flat[front_lit_flat]
            far, _ = intersect_circle(o, d, max_radius)
            n_far, n_near = far / 2.0, torch.ones_like(far) * 0.1
            near = torch.where(n_near < n_far, n_near, n_far)
            bg = torch.ones(1, 3) if use_white_bkgd else None
            ro = render(p_sdf, p_col, variance, cfg, o, d, near, far, max_radius, background_rgb=bg, cos_anneal_ratio=cos_anneal_ratio)
            occu = ro['weight_sum'].detach().numpy()
            front_lit_full = np.zeros(lvis_hit.shape, dtype=bool)
            front_lit_full[:, i:end_i] = front_lit.numpy()
            lvis_hit[front_lit_full] = 1.0 - occu[:, 0]
        batch_lvis.append(lvis_hit)
    lvis_hit = np.concatenate(batch_lvis, axis=0) if batch_lvis else np.zeros((0, n_lights), np.float32)
    lvis = np.zeros((surf.shape[0], n_lights), dtype=np.float32)
    lvis[alpha.numpy()] = lvis_hit
    return lvis
