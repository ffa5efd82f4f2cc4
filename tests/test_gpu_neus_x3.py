"""GPU: the exact-split NeuS kernels (vqn_neus_sdf_points_x3 / vqn_neus_fine_points_x3; every f32 operand as three bf16 pieces,
six bf16 MFMAs per product down to 2^-24, f32 accumulate) -- VERDICT r02 "next round" #3.

The gate for this engine is the F32 one, not the looser split-precision (f16s) one:
  * every comparison of tests/test_gpu_neus_mlp.py against the REAL reference's goldens and the oracle at the SAME tolerances the
    f32 kernels are held to (sdf 2e-5 abs, gradients 2e-4, rgb 2e-4) -- here; the render-level reference fixtures run under
    matrix_mode='x3' inside the f32 tests themselves, at their bounds (test_gpu_neus_render.py: test_render_core_all_keys,
    test_render_end_to_end_vs_reference; test_gpu_neus_hits.py: test_hits_render_variants_vs_reference);
  * error against the float64 evaluation of the same networks at the f32 kernels' own level: max |error| <= 1.25x theirs + 1e-7
    and mean |error| <= 1.25x theirs (round 4, activation pieces cut with round-to-nearest: sdf max 6.6e-7 vs 8.5e-7, mean 3.2e-7 vs
    3.1e-7; gradients max 5.9e-7 vs 1.12e-6, mean 1.2e-7 vs 2.0e-7; colour 6.5e-8 vs 6.8e-8, mean 1.8e-8 vs 1.9e-8.  Round 3, with
    truncated pieces: sdf max 9.7e-7, mean 5.3e-7 -- almost pure one-sided bias; the mean bound was 2x then);
  * no operand-range caveat: weights of 1e5 and activations of 1e-9 go through (the f16 pair engine refuses the former and loses the
    latter)."""
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
@pytest.mark.parametrize('P', [1, 33, 64, 65, 4099])
def test_sdf_points_x3(name, P):
    from oracle import geo as og
    from vqnerf_release_amd import _C
    from tests.gpu_util import launches
    cfg = _cfg(name)
    p_sdf, _, wb32, d32, _, _ = _packed(cfg, 'f32')
    _, _, wbx, dx, _, _ = _packed(cfg, 'x3')
    pts = torch.tensor(np.random.default_rng(3).uniform(-1.2, 1.2, (P, 3)).astype(np.float32))
    a = _C.neus_sdf_points(d32, wb32, pts=pts.cuda())
    with launches() as rec:
        b = _C.neus_sdf_points(dx, wbx, pts=pts.cuda(), mode='x3')
    assert rec.ran('vqn_neus_sdf_points_x3')
    with torch.no_grad():
        ref = og.sdf_only(p_sdf, cfg, pts)[:, 0]
    np.testing.assert_allclose(b.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)           # the f32 kernels' bound
    np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=0, atol=3e-6)
    # ray form (o + z d) = point form
    o = torch.zeros(P, 3).cuda()
    d = pts.cuda() / pts.cuda().norm(dim=-1, keepdim=True)
    z = pts.cuda().norm(dim=-1, keepdim=True).contiguous()
    c = _C.neus_sdf_points(dx, wbx, rays_o=o, rays_d=d.contiguous(), z=z, mode='x3')
    np.testing.assert_allclose(c.cpu().numpy(), _C.neus_sdf_points(d32, wb32, rays_o=o, rays_d=d.contiguous(), z=z).cpu().numpy(), rtol=0, atol=3e-6)
    # deterministic, and independent of where a point sits in its tile pair
    assert torch.equal(b, _C.neus_sdf_points(dx, wbx, pts=pts.cuda(), mode='x3'))
    if P > 40:
        perm = torch.randperm(P, generator=torch.Generator().manual_seed(0))
        bp = _C.neus_sdf_points(dx, wbx, pts=pts[perm].cuda(), mode='x3')
        assert torch.equal(bp.cpu(), b.cpu()[perm])


@pytest.mark.parametrize('name', ['full', 'small'])
def test_fine_points_x3_vs_golden_and_f32(name, golden_dir):
    from vqnerf_release_amd import _C
    cfg = _cfg(name)
    g = dict(np.load(os.path.join(golden_dir, f'geo_{name}.npz')))
    _, _, wb_s, d_s, wb_c, d_c = _packed(cfg, 'x3')
    _, _, wb_s32, d_s32, wb_c32, d_c32 = _packed(cfg, 'f32')
    rng = np.random.default_rng(3)
    pts = rng.uniform(-1.2, 1.2, (96, 3)).astype(np.float32)
    dirs = rng.normal(size=(96, 3)).astype(np.float32)
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    P, D = torch.tensor(pts).cuda(), torch.tensor(dirs).cuda()
    sdf, grad, rgb = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, pts=P, dirs=D, mode='x3')
    # the REAL reference's own outputs on these points, at the f32 kernels' tolerances (tests/test_gpu_neus_mlp.py)
    np.testing.assert_allclose(sdf.cpu().numpy(), g['net_sdf_out'][:, 0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), g['net_sdf_grad'], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rgb.cpu().numpy(), g['net_color'], rtol=0, atol=2e-4)
    # the f32 kernels
    s32, g32, c32 = _C.neus_fine_points(d_s32, wb_s32, d_c32, wb_c32, pts=P, dirs=D)
    np.testing.assert_allclose(sdf.cpu().numpy(), s32.cpu().numpy(), rtol=0, atol=3e-6)
    np.testing.assert_allclose(grad.cpu().numpy(), g32.cpu().numpy(), rtol=0, atol=5e-5 * float(g32.abs().max()))
    np.testing.assert_allclose(rgb.cpu().numpy(), c32.cpu().numpy(), rtol=0, atol=2e-5)
    assert not torch.equal(sdf, s32)                                 # (it is the other kernel)
    # the sdf of the fine kernel is the sdf-only kernel's, bit for bit (same packs, same order)
    assert torch.equal(_C.neus_sdf_points(d_s, wb_s, pts=P, mode='x3'), sdf)
    # gradient-only form (SDFNetwork.gradient) and ragged sizes
    zero_col = np.zeros_like(d_c)
    for n in (1, 31, 65):
        s2, g2, _ = _C.neus_fine_points(d_s, wb_s, zero_col, wb_c, pts=P[:n].contiguous(), dirs=D[:n].contiguous(), mode='x3')
        assert torch.equal(s2, sdf[:n]) and torch.equal(g2, grad[:n])


@pytest.mark.parametrize('name', ['full', 'small'])
def test_fine_points_x3_ray_mode_vs_oracle(name):
    from oracle import geo as og
    from vqnerf_release_amd import _C
    cfg = _cfg(name)
    p_sdf, p_col, wb_s, d_s, wb_c, d_c = _packed(cfg, 'x3')
    B, S = 37, 21                                  # ragged: P = 777 is not a multiple of 64
    o, d, near, far = map(torch.tensor, og.make_rays(B, 2))
    z = near + (far - near) * torch.linspace(0, 1, S)[None, :]
    sdf, grad, rgb = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, rays_o=o.cuda(), rays_d=d.cuda(), z=z.contiguous().cuda(), mode='x3')
    pts = (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3)
    dirs = d[:, None, :].expand(B, S, 3).reshape(-1, 3)
    with torch.no_grad():
        y = og.sdf_forward(p_sdf, cfg, pts)
    gr = og.sdf_gradient(p_sdf, cfg, pts)
    with torch.no_grad():
        c = og.color_forward(p_col, cfg, pts, gr, dirs, y[:, 1:])
    np.testing.assert_allclose(sdf.cpu().numpy(), y[:, 0].numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), gr.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rgb.cpu().numpy(), c.numpy(), rtol=0, atol=2e-4)


def test_x3_error_against_fp64_truth_is_not_above_the_f32_kernels():
    """Ground truth = the oracle in float64 (weights are the same fp32 numbers).  The gate of VERDICT r02 #3: the exact-split
    kernels' error budget <= the f32 kernels' (a 1.25x + 1e-7 allowance: two correct f32-accumulating evaluations that associate
    their sums differently do not have equal maxima over 2048 points)."""
    from oracle import geo as og
    from vqnerf_release_amd import _C
    cfg = og.FULL_CFG
    p_sdf, p_col, wb_s, d_s, wb_c, d_c = _packed(cfg, 'f32')
    _, _, wb_sx, d_sx, wb_cx, d_cx = _packed(cfg, 'x3')
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
    b = _C.neus_fine_points(d_sx, wb_sx, d_cx, wb_cx, pts=Pg, dirs=Dg, mode='x3')
    report = {}
    for name, x32, xx3, ref, floor in (('sdf', a[0], b[0], sdf64, 1e-7), ('grad', a[1], b[1], grad64, 2e-7), ('rgb', a[2], b[2], rgb64, 2e-8)):
        e32 = float((x32.double().cpu() - ref).abs().max())
        ex3 = float((xx3.double().cpu() - ref).abs().max())
        m32 = float((x32.double().cpu() - ref).abs().mean())
        mx3 = float((xx3.double().cpu() - ref).abs().mean())
        report[name] = (e32, ex3, m32, mx3)
    print('vs fp64 (max f32, max x3, mean f32, mean x3):', report)
    for name, floor in (('sdf', 1e-7), ('grad', 2e-7), ('rgb', 2e-8)):
        e32, ex3, m32, mx3 = report[name]
        assert ex3 <= 1.25 * e32 + floor, (name, e32, ex3)
        # mean |error|: sdf 3.2e-7 against the f32 kernels' 3.1e-7, gradients 1.2e-7 vs 2.0e-7, colour 1.8e-8 vs 1.9e-8 (round 3, with
        # truncated activation pieces: sdf 5.3e-7 -- the three cross terms an x3 product drops all carried the product's sign)
        assert mx3 <= 1.25 * m32 + floor / 10, (name, m32, mx3)


def test_x3_has_no_operand_range_caveat():
    """Weights of 1e5 (the f16 pair packs refuse |w| > 6e4) and inputs that make activations of ~1e-9 (below what an f16 pair
    resolves): the exact split carries both; results match the f32 kernels at relative 1e-5 of the output scale."""
    from vqnerf_release_amd import _C
    from vqnerf_release_amd.geo.models.fields import SDFNetwork
    torch.manual_seed(3)
    sdf = SDFNetwork(d_in=3, d_out=65, d_hidden=64, n_layers=4, skip_in=(2,), multires=6, weight_norm=False, geometric_init=False).cuda()
    with torch.no_grad():
        sdf.lin0.weight.mul_(1e-3)
        sdf.lin1.weight.mul_(2e5 / float(sdf.lin1.weight.abs().max()))          # |w| up to 2e5
        sdf.lin1.bias.zero_()
        sdf.lin2.weight.mul_(1e-7)
    pts = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (200, 3)).astype(np.float32)).cuda()
    wb32, d32 = sdf.packs(max_tiles=2, mode='f32')
    wbx, dx = sdf.packs(max_tiles=2, mode='x3')
    with pytest.raises(ValueError):
        sdf.packs(max_tiles=2, mode='f16s')
    a = _C.neus_sdf_points(d32, wb32, pts=pts)
    b = _C.neus_sdf_points(dx, wbx, pts=pts, mode='x3')
    assert torch.isfinite(b).all()
    np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-5, atol=1e-5 * float(a.abs().max()))


def test_x3_refuses_layers_wider_than_eight_tiles():
    from vqnerf_release_amd import _C
    from vqnerf_release_amd.geo.models.fields import SDFNetwork
    sdf = SDFNetwork(d_in=3, d_out=257, d_hidden=320, n_layers=4, skip_in=(2,), multires=6).cuda()
    wbx, dx = sdf.packs(max_tiles=10, mode='x3')
    pts = torch.zeros(8, 3).cuda()
    with pytest.raises(_C.VqnError, match='x3'):
        _C.neus_sdf_points(dx, wbx, pts=pts, mode='x3')


SHAPES = [
    # d_hidden, n_layers, skip_in, multires, col_hidden, col_layers, multires_view
    (64, 4, (2,), 6, 64, 2, 4),        # 2-tile layers, 25-wide layer before the skip (one partial tile), odd step counts
    (96, 3, (), 4, 64, 1, 2),          # no skip, 3-tile layers, 27-feature embedding (2 steps), one colour layer
    (256, 8, (4,), 6, 256, 4, 4),      # the shipped shape
    (224, 5, (1,), 6, 128, 3, 4),      # 7-tile layers, skip right after the first layer
    (160, 2, (), 10, 96, 2, 1),        # 63-feature embedding (4 steps, the E region full), 5-tile layers
    (32, 6, (3,), 2, 32, 2, 2),        # one-tile layers: seven of the eight waves idle
]


@pytest.mark.parametrize('shape', SHAPES, ids=[f'h{s[0]}l{s[1]}' for s in SHAPES])
def test_x3_network_shapes_vs_f32_kernels(shape):
    """The exact-split kernels on network shapes other than the shipped one -- partial tiles, odd numbers of K steps (zero-padded
    blocks), embeddings of 1 / 2 / 4 steps, one-tile layers, no skip / early skip -- against the f32 kernels (which
    tests/test_gpu_neus_mlp.py holds to the torch statement): sdf, d sdf / dx and colour at f32-level differences, at ragged P."""
    from vqnerf_release_amd import _C
    from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork
    dh, nl, skip, mr, ch, cl, mrv = shape
    torch.manual_seed(dh + nl)
    sdf = SDFNetwork(d_in=3, d_out=dh + 1, d_hidden=dh, n_layers=nl, skip_in=skip, multires=mr, bias=0.5, scale=1.0,
                     geometric_init=True, weight_norm=True).cuda()
    col = RenderingNetwork(d_feature=dh, mode='idr', d_in=9, d_out=3, d_hidden=ch, n_layers=cl, weight_norm=True,
                           multires_view=mrv, squeeze_out=True).cuda()
    with torch.no_grad():
        for p_ in list(sdf.parameters()) + list(col.parameters()):
            p_.add_(0.02 * torch.randn_like(p_))
    rng = np.random.default_rng(nl)
    for P in (1, 77, 1000):
        pts = torch.tensor(rng.uniform(-1, 1, (P, 3)).astype(np.float32)).cuda()
        dirs = torch.nn.functional.normalize(torch.tensor(rng.normal(size=(P, 3)).astype(np.float32)), dim=-1).cuda()
        out = {}
        for mode in ('f32', 'x3'):
            wb_s, d_s = sdf.packs(max_tiles=col.max_tiles(), mode=mode)
            wb_c, d_c = col.packs(feat_tiles=sdf.plan(mode=mode).tiles[-1], mode=mode)
            out[mode] = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, pts=pts, dirs=dirs, mode=mode) + (_C.neus_sdf_points(d_s, wb_s, pts=pts, mode=mode),)
        (s32, g32, c32, o32), (sx, gx, cx, ox) = out['f32'], out['x3']
        assert torch.isfinite(sx).all() and torch.isfinite(gx).all() and torch.isfinite(cx).all()
        np.testing.assert_allclose(sx.cpu().numpy(), s32.cpu().numpy(), rtol=0, atol=3e-6 * max(1.0, float(s32.abs().max())))
        np.testing.assert_allclose(gx.cpu().numpy(), g32.cpu().numpy(), rtol=0, atol=5e-5 * max(1.0, float(g32.abs().max())))
        np.testing.assert_allclose(cx.cpu().numpy(), c32.cpu().numpy(), rtol=0, atol=2e-5)
        assert torch.equal(ox, sx)                              # the SDF-only kernel = the fine kernel's sdf, bit for bit
