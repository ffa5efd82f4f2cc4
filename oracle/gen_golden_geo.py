#!/usr/bin/env python
"""Generate tests/golden/geo_*.npz by RUNNING THE REAL REFERENCE (geo half).

Build-container only: imports /root/reference/geo/NeuS-ours2/models/{renderer,fields}.py
(with empty stand-in modules for its two unused top-level imports `mcubes` and
`icecream`), feeds it numpy-seeded weights/rays (oracle.geo.make_*), and stores
ONLY the reference's outputs.  Inputs are regenerated from the same seeds by
the tests, so the fixtures stay small.  The reference source never ships.

    python oracle/gen_golden_geo.py            # writes tests/golden/geo_*.npz
"""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import geo as og  # noqa: E402

REF = os.environ.get('VQNERF_REFERENCE', '/root/reference')
GOLD = os.path.join(ROOT, 'tests', 'golden')


def import_reference():
    for m in ('mcubes', 'icecream'):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.modules['icecream'].ic = lambda *a, **k: None
    sys.path.insert(0, os.path.join(REF, 'geo', 'NeuS-ours2'))
    import warnings
    warnings.filterwarnings('ignore')
    from models import renderer as R, fields as Fd
    return R, Fd


def build_ref(Fd, R, cfg, p_sdf, p_col, variance):
    c = cfg['sdf']
    sdf = Fd.SDFNetwork(d_in=c['d_in'], d_out=c['d_out'], d_hidden=c['d_hidden'], n_layers=c['n_layers'],
                        skip_in=tuple(c['skip_in']), multires=c['multires'], bias=c['bias'], scale=c['scale'],
                        geometric_init=True, weight_norm=True)
    sdf.load_state_dict({k: torch.tensor(v) for k, v in p_sdf.items()})
    cc = cfg['color']
    col = Fd.RenderingNetwork(d_feature=cc['d_feature'], mode=cc['mode'], d_in=cc['d_in'], d_out=cc['d_out'],
                              d_hidden=cc['d_hidden'], n_layers=cc['n_layers'], weight_norm=True,
                              multires_view=cc['multires_view'], squeeze_out=cc['squeeze_out'])
    col.load_state_dict({k: torch.tensor(v) for k, v in p_col.items()})
    var = Fd.SingleVarianceNetwork(variance)
    ren = R.NeuSRenderer(None, sdf, var, col, **cfg['renderer'])
    return sdf, col, var, ren


def np_(d):
    out = {}
    for k, v in d.items():
        if torch.is_tensor(v):
            out[k] = v.detach().cpu().numpy()
        elif isinstance(v, (float, int, bool, np.ndarray)):
            out[k] = np.asarray(v)
    return out


HITS_VARIANCE = 0.5        # inv_s = e^5 ~ 148: surface hits composite to weight_sum ~ 1 (a trained net's regime)


def gen_hits(R, Fd):
    """tests/golden/geo_hits.npz -- full-size nets, 64 rays of which > 1/3 reach weight_sum > 0.9, 8 miss the bounding
    sphere entirely: (a) render, all keys; (b) render with perturb = 1 and the jitter INJECTED (the reference draws
    `torch.rand([B,1])` at renderer.py:318; `torch.rand` is swapped for the duration of that one call so that the reference runs
    on our array); (c) render(to_light=True) with per-ray near / far (renderer.py:302, :211); (d) the up-sampling stages on
    these rays; (e) gradients of the training loss on these rays."""
    cfg = og.FULL_CFG
    p_sdf, p_col = og.make_sdf_params(cfg, seed=0), og.make_color_params(cfg, seed=1)
    sdf, col, var, ren = build_ref(Fd, R, cfg, p_sdf, p_col, HITS_VARIANCE)
    rays = og.make_hit_rays()
    t = {k: torch.tensor(v) for k, v in rays.items()}
    B, radius, white = len(rays['o']), 2.0, torch.ones(1, 3)
    out = {}
    rr = ren.render(t['o'], t['d'], t['near'], t['far'], radius, perturb_overwrite=0, background_rgb=white, cos_anneal_ratio=1.0)
    for k, v in np_(rr).items():
        out[f'render_{k}'] = v
    ws = out['render_weight_sum'].ravel()
    assert (ws > 0.9).mean() >= 1 / 3 and (ws < 0.1).mean() >= 1 / 4, ((ws > 0.9).mean(), (ws < 0.1).mean())
    assert out['render_inside_sphere'][-8:].max() == 0.0 and out['render_inside_sphere'][:-8].max() == 1.0
    rr = ren.render(t['o'], t['d'], t['near'], t['far'], radius, perturb_overwrite=0, background_rgb=None, cos_anneal_ratio=0.5)
    for k, v in np_(rr).items():
        out[f'render_none0.5_{k}'] = v
    # (b) injected jitter
    real_rand = torch.rand
    calls = []

    def fake_rand(shape, *a, **k):
        assert list(shape) == [B, 1], shape
        calls.append(1)
        return t['t_rand'] + 0.5
    torch.rand = fake_rand
    try:
        rr = ren.render(t['o'], t['d'], t['near'], t['far'], radius, perturb_overwrite=1, background_rgb=white, cos_anneal_ratio=1.0)
    finally:
        torch.rand = real_rand
    assert len(calls) == 1
    for k, v in np_(rr).items():
        out[f'perturb_{k}'] = v
    # (c) to_light
    rr = ren.render(t['o'], t['d'], t['near_l'], t['far_l'], radius, perturb_overwrite=0, background_rgb=white,
                    cos_anneal_ratio=1.0, to_light=True)
    for k, v in np_(rr).items():
        out[f'tolight_{k}'] = v
    # (d) up-sampling stages
    n0 = cfg['renderer']['n_samples']
    z_vals = t['near'] + (t['far'] - t['near']) * torch.linspace(0.0, 1.0, n0)[None, :]
    with torch.no_grad():
        ptsz = t['o'][:, None, :] + t['d'][:, None, :] * z_vals[..., None]
        sd = sdf.sdf(ptsz.reshape(-1, 3)).reshape(B, n0)
        out['coarse_sdf'] = sd.numpy()
        zz, ss = z_vals, sd
        for i in range(4):
            new_z = ren.up_sample(t['o'], t['d'], zz, ss, radius, 16, 64 * 2 ** i)
            out[f'up_new_z_{i}'] = new_z.numpy()
            zz, ss = ren.cat_z_vals(t['o'], t['d'], zz, new_z, ss, last=(i == 3))
            out[f'up_z_{i}'] = zz.numpy()
            out[f'up_sdf_{i}'] = ss.numpy()
            out[f'up_ties_{i}'] = np.asarray(bool((zz[:, 1:] == zz[:, :-1]).any()))
    # render_core on exactly these depths (stage-isolated: end-to-end, the depths of samples on near-empty rays are
    # ill-conditioned -- their pdf is (w + 1e-5) / sum with w ~ 1e-5 -- so per-sample keys are pinned here instead)
    rc = ren.render_core(t['o'], t['d'], zz, 2 * radius / n0, radius, sdf, var, col, background_rgb=white, cos_anneal_ratio=1.0)
    for k, v in np_(rc).items():
        out[f'core_{k}'] = v
    rc = ren.render_core(t['o'], t['d'], zz, (t['far_l'] - t['near_l']) / n0, radius, sdf, var, col, background_rgb=white,
                         cos_anneal_ratio=0.5, to_light=True)
    for k, v in np_(rc).items():
        out[f'coretl_{k}'] = v
    # (e) backward
    for m in (sdf, col, var):
        m.zero_grad()
    rr = ren.render(t['o'], t['d'], t['near'], t['far'], radius, perturb_overwrite=0, background_rgb=white, cos_anneal_ratio=1.0)
    tgt = torch.tensor(np.random.default_rng(4).uniform(0, 1, (B, 3)).astype(np.float32))
    loss = (rr['color_fine'] - tgt).abs().sum() / B + 0.1 * rr['gradient_error']
    loss.backward()
    out['bwd_loss'] = loss.detach().numpy()
    for name, m in (('sdf', sdf), ('col', col), ('var', var)):
        for k, prm in m.named_parameters():
            out[f'bwd_{name}.{k}'] = prm.grad.numpy()
    np.savez_compressed(os.path.join(GOLD, 'geo_hits.npz'), **out)
    print('hits keys:', len(out), 'weight_sum > 0.9:', float((ws > 0.9).mean()), '< 0.1:', float((ws < 0.1).mean()),
          'ties:', [bool(out[f'up_ties_{i}']) for i in range(4)])


def gen_upsample_edge(R, Fd):
    """tests/golden/geo_upsample_edge.npz -- the reference's `up_sample` (and through it `sample_pdf`) on hand-made SDF
    profiles (oracle.geo.make_upsample_edge_inputs): the edge branches of both functions in the form the fused HIP kernel
    takes its inputs (depths + SDF values, not bins + weights)."""
    cfg = og.SMALL_CFG
    sdf, col, var, ren = build_ref(Fd, R, cfg, og.make_sdf_params(cfg, 0), og.make_color_params(cfg, 1), 0.3)
    out = {}
    for i, n in enumerate((64, 80, 96, 112)):
        o, d, z, s = map(torch.tensor, og.make_upsample_edge_inputs(n))
        with torch.no_grad():
            out[f'new_z_{n}'] = ren.up_sample(o, d, z, s, 2.0, 16, 64 * 2 ** i).numpy()
        assert np.isfinite(out[f'new_z_{n}']).all()
    np.savez_compressed(os.path.join(GOLD, 'geo_upsample_edge.npz'), **out)
    print('upsample_edge keys:', len(out))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    R, Fd = import_reference()

    # ---------------- (i) sample_pdf, det=True ----------------
    rng = np.random.default_rng(10)
    out = {}
    for n in (64, 80, 96, 112):
        bins = np.sort(rng.uniform(2.0, 6.0, (8, n)).astype(np.float32), -1)
        w = rng.uniform(0, 1, (8, n - 1)).astype(np.float32) ** 4
        w[0] = 0.0                      # all-zero weights row
        w[1, : n // 2] = 0.0            # half-empty
        w[2] = 0.0; w[2, 5] = 1.0       # single spike -> denom<1e-5 branches elsewhere
        s = R.sample_pdf(torch.tensor(bins), torch.tensor(w), 16, det=True)
        out[f'samples_{n}'] = s.numpy()
    np.savez_compressed(os.path.join(GOLD, 'geo_sample_pdf.npz'), **out)

    for tag, cfg, B in (('full', og.FULL_CFG, 16), ('small', og.SMALL_CFG, 64)):
        p_sdf = og.make_sdf_params(cfg, seed=0)
        p_col = og.make_color_params(cfg, seed=1)
        variance = 0.3
        sdf, col, var, ren = build_ref(Fd, R, cfg, p_sdf, p_col, variance)
        o, d, near, far = og.make_rays(B, seed=2)
        o_t, d_t, near_t, far_t = map(torch.tensor, (o, d, near, far))
        radius = 2.0
        out = {}

        # ---------------- networks ----------------
        rng = np.random.default_rng(3)
        pts = rng.uniform(-1.2, 1.2, (96, 3)).astype(np.float32)
        dirs = rng.normal(size=(96, 3)).astype(np.float32)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        with torch.no_grad():
            y = sdf(torch.tensor(pts))
        out['net_sdf_out'] = y.numpy()
        g = sdf.gradient(torch.tensor(pts)).squeeze().detach()          # (vi)
        out['net_sdf_grad'] = g.numpy()
        with torch.no_grad():
            rgb = col(torch.tensor(pts), g, torch.tensor(dirs), y[:, 1:])
        out['net_color'] = rgb.numpy()

        # ---------------- (ii)/(iii) up_sample + cat_z_vals, each inv_s ----------------
        n0 = cfg['renderer']['n_samples']
        z = torch.linspace(0.0, 1.0, n0)
        z_vals = near_t + (far_t - near_t) * z[None, :]
        with torch.no_grad():
            ptsz = o_t[:, None, :] + d_t[:, None, :] * z_vals[..., None]
            sd = sdf.sdf(ptsz.reshape(-1, 3)).reshape(B, n0)
            out['coarse_sdf'] = sd.numpy()
            zz, ss = z_vals, sd
            for i in range(4):
                new_z = ren.up_sample(o_t, d_t, zz, ss, radius, 16, 64 * 2 ** i)
                out[f'up_new_z_{i}'] = new_z.numpy()
                zz, ss = ren.cat_z_vals(o_t, d_t, zz, new_z, ss, last=(i == 3))
                out[f'up_z_{i}'] = zz.numpy()
                out[f'up_sdf_{i}'] = ss.numpy()
                out[f'up_ties_{i}'] = np.asarray(bool((zz[:, 1:] == zz[:, :-1]).any()))

        # ---------------- (iv) render_core, all keys ----------------
        z_fine = zz if cfg['renderer']['n_importance'] > 0 else z_vals
        n_core = z_fine.shape[1]
        for car in (0.0, 0.5, 1.0):
            rc = ren.render_core(o_t, d_t, z_fine, 2 * radius / n0, radius, sdf, var, col,
                                 background_rgb=torch.ones(1, 3), cos_anneal_ratio=car)
            for k, v in np_(rc).items():
                out[f'core{car}_{k}'] = v
        out['core_z_in'] = z_fine.numpy()

        # ---------------- (v) render end to end ----------------
        for bg_name, bg in (('white', torch.ones(1, 3)), ('none', None)):
            for car in (0.0, 0.5, 1.0):
                rr = ren.render(o_t, d_t, near_t, far_t, radius, perturb_overwrite=0,
                                background_rgb=bg, cos_anneal_ratio=car)
                for k, v in np_(rr).items():
                    out[f'render_{bg_name}_{car}_{k}'] = v

        # ---------------- (vii) backward ----------------
        if tag == 'full' or True:
            for m in (sdf, col, var):
                m.zero_grad()
            rr = ren.render(o_t, d_t, near_t, far_t, radius, perturb_overwrite=0,
                            background_rgb=torch.ones(1, 3), cos_anneal_ratio=1.0)
            tgt = torch.tensor(np.random.default_rng(4).uniform(0, 1, (B, 3)).astype(np.float32))
            loss = (rr['color_fine'] - tgt).abs().sum() / B + 0.1 * rr['gradient_error']
            loss.backward()
            out['bwd_loss'] = loss.detach().numpy()
            for name, m in (('sdf', sdf), ('col', col), ('var', var)):
                for k, prm in m.named_parameters():
                    out[f'bwd_{name}.{k}'] = prm.grad.numpy()

        np.savez_compressed(os.path.join(GOLD, f'geo_{tag}.npz'), **out)
        print(tag, 'keys:', len(out), 'ties:', [bool(out[f'up_ties_{i}']) for i in range(4)])

    gen_hits(R, Fd)
    gen_upsample_edge(R, Fd)

    # ---------------- gen_light_xyz (pure-numpy twin at geo/models/util.py:84-119) ----------------
    from models import util as U
    xyz, areas = U.gen_light_xyz(16, 32)
    np.savez_compressed(os.path.join(GOLD, 'light_xyz_16x32.npz'), xyz=xyz, areas=areas)
    print('done ->', GOLD)


if __name__ == '__main__':
    main()
