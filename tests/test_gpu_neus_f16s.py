"""GPU: the split-precision NeuS kernels (vqn_neus_sdf_points_f16s / vqn_neus_fine_points_f16s; f16 hi/lo operands, three f16
MFMAs per product, f32 accumulate) -- an OPT-IN mode with a stated tolerance instead of bitwise agreement.

Stated tolerance: the same bounds the f32 kernels are held to against the reference goldens and the oracle (sdf 2e-5 abs,
gradients 2e-4, rgb 2e-4), and against the f32 kernels themselves sdf 5e-6, gradients 1e-4 relative to the gradient scale,
rgb 5e-5; a rendered image within 80 dB PSNR of the f32 render."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(name):
    from oracle import geo as og
    return og.FULL_CFG if name == 'full' else og.SMALL_CFG


def _packed(cfg, mode, dev='cuda'):
    from oracle import geo as og
    from vqnerf_release_amd.geo import packing as pk
    c, cc = cfg['sdf'], cfg['color']
    sp = pk.SdfPackPlan(og.sdf_dims(cfg), c['skip_in'], c['multires'], c['scale'], max_tiles=(cc['d_hidden'] + 31) // 32, mode=mode)
    cp = pk.ColPackPlan(cc['d_feature'], cc['mode'], cc['d_hidden'], cc['n_layers'], cc['d_out'], cc['multires_view'],
                        cc['squeeze_out'], feat_tiles=sp.tiles[-1], matrix_mode=mode)
    p_sdf = og.to_torch(og.make_sdf_params(cfg, 0))
    p_col = og.to_torch(og.make_color_params(cfg, 1))
    Ws = [og.wn_weight(p_sdf, l).to(dev) for l in range(sp.n_lin)]
    bs = [p_sdf[f'lin{l}.bias'].to(dev) for l in range(sp.n_lin)]
    Wc = [og.wn_weight(p_col, l).to(dev) for l in range(cp.n_lin)]
    bc = [p_col[f'lin{l}.bias'].to(dev) for l in range(cp.n_lin)]
    return (p_sdf, p_col) + sp.pack(Ws, bs) + cp.pack(Wc, bc)


@pytest.mark.parametrize('name', ['full', 'small'])
@pytest.mark.parametrize('P', [1, 33, 4099])
def test_sdf_points_f16s(name, P):
    from oracle import geo as og
    from vqnerf_release_amd import _C
    cfg = _cfg(name)
    p_sdf, _, wb32, d32, _, _ = _packed(cfg, 'f32')
    _, _, wb16, d16, _, _ = _packed(cfg, 'f16s')
    pts = torch.tensor(np.random.default_rng(3).uniform(-1.2, 1.2, (P, 3)).astype(np.float32))
    a = _C.neus_sdf_points(d32, wb32, pts=pts.cuda())
    b = _C.neus_sdf_points(d16, wb16, pts=pts.cuda(), mode='f16s')
    with torch.no_grad():
        ref = og.sdf_only(p_sdf, cfg, pts)[:, 0]
    np.testing.assert_allclose(b.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=0, atol=5e-6)
    # ray form (o + z d) = point form
    o = torch.zeros(P, 3).cuda()
    d = pts.cuda() / pts.cuda().norm(dim=-1, keepdim=True)
    z = pts.cuda().norm(dim=-1, keepdim=True).contiguous()
    c = _C.neus_sdf_points(d16, wb16, rays_o=o, rays_d=d.contiguous(), z=z, mode='f16s')
    np.testing.assert_allclose(c.cpu().numpy(), _C.neus_sdf_points(d32, wb32, rays_o=o, rays_d=d.contiguous(), z=z).cpu().numpy(), rtol=0, atol=5e-6)


@pytest.mark.parametrize('name', ['full', 'small'])
def test_fine_points_f16s_vs_golden_and_f32(name, golden_dir):
    from vqnerf_release_amd import _C
    cfg = _cfg(name)
    g = dict(np.load(os.path.join(golden_dir, f'geo_{name}.npz')))
    _, _, wb_s, d_s, wb_c, d_c = _packed(cfg, 'f16s')
    _, _, wb_s32, d_s32, wb_c32, d_c32 = _packed(cfg, 'f32')
    rng = np.random.default_rng(3)
    pts = rng.uniform(-1.2, 1.2, (96, 3)).astype(np.float32)
    dirs = rng.normal(size=(96, 3)).astype(np.float32)
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    P, D = torch.tensor(pts).cuda(), torch.tensor(dirs).cuda()
    sdf, grad, rgb = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, pts=P, dirs=D, mode='f16s')
    # the reference's own outputs on these points
    np.testing.assert_allclose(sdf.cpu().numpy(), g['net_sdf_out'][:, 0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), g['net_sdf_grad'], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rgb.cpu().numpy(), g['net_color'], rtol=0, atol=2e-4)
    # the f32 kernels
    s32, g32, c32 = _C.neus_fine_points(d_s32, wb_s32, d_c32, wb_c32, pts=P, dirs=D)
    np.testing.assert_allclose(sdf.cpu().numpy(), s32.cpu().numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(grad.cpu().numpy(), g32.cpu().numpy(), rtol=0, atol=1e-4 * float(g32.abs().max()))
    np.testing.assert_allclose(rgb.cpu().numpy(), c32.cpu().numpy(), rtol=0, atol=5e-5)
    assert not torch.equal(sdf, s32)                                 # (it is the other kernel)
    # gradient-only form (SDFNetwork.gradient) and ragged sizes
    zero_col = np.zeros_like(d_c)
    for n in (1, 31, 65):
        s2, g2, _ = _C.neus_fine_points(d_s, wb_s, zero_col, wb_c, pts=P[:n].contiguous(), dirs=D[:n].contiguous(), mode='f16s')
        assert torch.equal(s2, sdf[:n]) and torch.equal(g2, grad[:n])


def test_render_f16s_matches_f32_render():
    """matrix_mode='f16s' through NeuSRenderer.render (up-sampling on the f16s SDF kernel, fine pass on the f16s fine kernel)."""
    from oracle import geo as og
    from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork
    from vqnerf_release_amd.geo.models.renderer import NeuSRenderer
    cfg = og.FULL_CFG
    torch.manual_seed(0)
    sdf = SDFNetwork(**cfg['sdf']).cuda()
    col = RenderingNetwork(**cfg['color']).cuda()
    var = SingleVarianceNetwork(0.3).cuda()
    ren = NeuSRenderer(None, sdf, var, col, n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=0.0)
    B = 600
    rng = np.random.default_rng(5)
    d = rng.normal(size=(B, 3)).astype(np.float32) * 0.15 + np.array([0, 0, -1], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(np.array([[0, 0, 4.0]], np.float32), (B, 1))
    O, D = torch.tensor(o).cuda(), torch.tensor(d).cuda()
    near, far = torch.full((B, 1), 2.0).cuda(), torch.full((B, 1), 6.0).cuda()
    with torch.no_grad():
        a = ren.render(O, D, near, far, 2.0, perturb_overwrite=0, background_rgb=torch.ones(1, 3).cuda(), cos_anneal_ratio=1.0)
        ren.matrix_mode = 'f16s'
        b = ren.render(O, D, near, far, 2.0, perturb_overwrite=0, background_rgb=torch.ones(1, 3).cuda(), cos_anneal_ratio=1.0)
    mse = float(((a['color_fine'] - b['color_fine']) ** 2).mean())
    assert -10 * np.log10(mse + 1e-30) > 80.0, mse
    assert float(a['weight_sum'].max()) > 0.5                        # rays do hit the surface
    np.testing.assert_allclose(b['weight_sum'].cpu().numpy(), a['weight_sum'].cpu().numpy(), rtol=0, atol=2e-4)
    # per-sample gradients: the sample positions themselves move by ~1e-6 (importance sampling on the other SDF kernel)
    dg = (b['gradients'] - a['gradients']).abs().cpu().numpy()
    assert np.quantile(dg, 0.999) < 1e-3 and dg.max() < 2e-2, (np.quantile(dg, 0.999), dg.max())


def test_wide_network_takes_the_one_image_kernel():
    """d_hidden = 320 (10 tiles): two 32-point images no longer fit in LDS, the entry points fall back to the one-image
    workgroup form -- same results against the f32 kernels."""
    from vqnerf_release_amd import _C
    from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork
    torch.manual_seed(1)
    sdf = SDFNetwork(d_in=3, d_out=257, d_hidden=320, n_layers=4, skip_in=(2,), multires=6, bias=0.5, scale=1.0,
                     geometric_init=True, weight_norm=True).cuda()
    col = RenderingNetwork(d_feature=256, mode='idr', d_in=9, d_out=3, d_hidden=320, n_layers=2, weight_norm=True,
                           multires_view=4, squeeze_out=True).cuda()
    rng = np.random.default_rng(9)
    P = torch.tensor(rng.uniform(-1, 1, (77, 3)).astype(np.float32)).cuda()
    D = torch.nn.functional.normalize(torch.tensor(rng.normal(size=(77, 3)).astype(np.float32)), dim=-1).cuda()
    out = {}
    for mode in ('f32', 'f16s'):
        wb_s, d_s = sdf.packs(max_tiles=col.max_tiles(), mode=mode)
        wb_c, d_c = col.packs(feat_tiles=sdf.plan(mode=mode).tiles[-1], mode=mode)
        out[mode] = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, pts=P, dirs=D, mode=mode) + (_C.neus_sdf_points(d_s, wb_s, pts=P, mode=mode),)
    for a, b, tol in zip(out['f16s'], out['f32'], (5e-6, None, 5e-5, 5e-6)):
        tol = tol if tol is not None else 1e-4 * float(b.abs().max())
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=tol)


def test_f16s_error_against_fp64_truth_is_fp32_level():
    """Ground truth = the oracle evaluated in float64 (weights are the same fp32 numbers).  The split-precision kernels must be
    as close to it as the exact-f32 kernels are, up to a small factor: sdf and colour within 3x the f32 kernels' own error
    (+ 1e-7 floor), gradients within 3x (+ 1e-6).  Measured on MI355X: ratios 1.0-1.6."""
    from oracle import geo as og
    from vqnerf_release_amd import _C
    cfg = og.FULL_CFG
    p_sdf, p_col, wb_s, d_s, wb_c, d_c = _packed(cfg, 'f32')
    _, _, wb_s16, d_s16, wb_c16, d_c16 = _packed(cfg, 'f16s')
    rng = np.random.default_rng(11)
    n = 2048
    pts = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    P64, D64 = torch.tensor(pts, dtype=torch.float64), torch.tensor(dirs, dtype=torch.float64)
    p64 = {k: v.double() for k, v in p_sdf.items()}
    c64 = {k: v.double() for k, v in p_col.items()}
    with torch.no_grad():
        out64 = og.sdf_forward(p64, cfg, P64)
    sdf64, feat64 = out64[:, 0], out64[:, 1:]
    grad64 = og.sdf_gradient(p64, cfg, P64)
    with torch.no_grad():
        rgb64 = og.color_forward(c64, cfg, P64, grad64, D64, feat64)
    Pg, Dg = torch.tensor(pts).cuda(), torch.tensor(dirs).cuda()
    a = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, pts=Pg, dirs=Dg)
    b = _C.neus_fine_points(d_s16, wb_s16, d_c16, wb_c16, pts=Pg, dirs=Dg, mode='f16s')
    report = {}
    for name, x32, x16, ref, floor in (('sdf', a[0], b[0], sdf64, 1e-7), ('grad', a[1], b[1], grad64, 1e-6), ('rgb', a[2], b[2], rgb64, 1e-7)):
        e32 = float((x32.double().cpu() - ref).abs().max())
        e16 = float((x16.double().cpu() - ref).abs().max())
        report[name] = (e32, e16)
        assert e16 <= 3.0 * e32 + floor, (name, e32, e16)
    print('max |error| vs fp64 (f32 kernel, f16s kernel):', report)
