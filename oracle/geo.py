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
    """renderer.py:177-191.  Returns (z_sorted, sdf_sorted, had_ties)."""
    B, n = z_vals.shape
    m = new_z.shape[1]
    z_cat = torch.cat([z_vals, new_z], -1)
    z_sorted, index = torch.sort(z_cat, dim=-1)
    ties = bool((z_sorted[:, 1:] == z_sorted[:, :-1]).any())
    if not last:
        pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z[..., None]
        new_sdf = sdf_only(p_sdf, cfg, pts.reshape(-1, 3)).reshape(B, m)
        sdf = torch.gather(torch.cat([sdf, new_sdf], -1), -1, index)
    return z_sorted, sdf, ties


def render_core(p_sdf, p_col, variance, cfg, rays_o, rays_d, z_vals, sample_dist, radius,
                background_rgb=None, cos_anneal_ratio=0.0, create_graph=False):
    """renderer.py:193-297 (n_outside == 0 branch)."""
    B, n = z_vals.shape
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], float(sample_dist))], -1)
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
           background_rgb=None, cos_anneal_ratio=0.0, create_graph=False):
    """renderer.py:299-401 (n_outside == 0, to_light=False).

    `jitter` is the explicit stand-in for ``torch.rand([B,1]) - 0.5`` at
    renderer.py:318 (None == perturb 0)."""
    r = cfg['renderer']
    B = rays_o.shape[0]
    sample_dist = 2.0 * radius / r['n_samples']
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
                     create_graph=create_graph)
    w = rc['weights']
    return dict(color_fine=rc['color'],
                s_val=rc['s_val'].reshape(B, n).mean(-1, keepdim=True),
                cdf_fine=rc['cdf'], weight_sum=w.sum(-1, keepdim=True),
                weight_max=w.max(-1, keepdim=True)[0], gradients=rc['gradients'],
                weights=w, gradient_error=rc['gradient_error'],
                inside_sphere=rc['inside_sphere'], surf=rc['surf'], depth=rc['depth'],
                z_vals=z_vals, had_ties=ties)
