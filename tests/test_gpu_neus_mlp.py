"""GPU parity: the fused NeuS network kernels against the geo oracle (oracle/geo.py, which is
pinned to the real reference by tests/golden/geo_*.npz) and against the goldens directly.

fp32 tolerance: |sdf| 2e-5 abs, gradients 2e-4, rgb 2e-4 (the kernel sums each dot product in a
different -- fixed -- order than BLAS and uses v_exp/v_log for softplus)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _plans(cfg):
    from oracle import geo as og
    from vqnerf_release_amd.geo import packing as pk
    c, cc = cfg['sdf'], cfg['color']
    sp = pk.SdfPackPlan(og.sdf_dims(cfg), c['skip_in'], c['multires'], c['scale'],
                        max_tiles=(cc['d_hidden'] + 31) // 32)
    cp = pk.ColPackPlan(cc['d_feature'], cc['mode'], cc['d_hidden'], cc['n_layers'], cc['d_out'],
                        cc['multires_view'], cc['squeeze_out'], feat_tiles=sp.tiles[-1])
    return sp, cp


def _packed(cfg, dev='cuda'):
    from oracle import geo as og
    p_sdf = og.to_torch(og.make_sdf_params(cfg, 0))
    p_col = og.to_torch(og.make_color_params(cfg, 1))
    sp, cp = _plans(cfg)
    Ws = [og.wn_weight(p_sdf, l).to(dev) for l in range(sp.n_lin)]
    bs = [p_sdf[f'lin{l}.bias'].to(dev) for l in range(sp.n_lin)]
    Wc = [og.wn_weight(p_col, l).to(dev) for l in range(cp.n_lin)]
    bc = [p_col[f'lin{l}.bias'].to(dev) for l in range(cp.n_lin)]
    wb_s, d_s = sp.pack(Ws, bs)
    wb_c, d_c = cp.pack(Wc, bc)
    return p_sdf, p_col, wb_s, d_s, wb_c, d_c


CFGS = ['full', 'small']


def _cfg(name):
    from oracle import geo as og
    return og.FULL_CFG if name == 'full' else og.SMALL_CFG


@pytest.mark.parametrize('name', CFGS)
@pytest.mark.parametrize('P', [96, 1, 33, 4099])
def test_sdf_points_vs_oracle(name, P):
    from oracle import geo as og
    from vqnerf_release_amd import _C
    cfg = _cfg(name)
    p_sdf, _, wb_s, d_s, _, _ = _packed(cfg)
    rng = np.random.default_rng(3)
    pts = torch.tensor(rng.uniform(-1.2, 1.2, (P, 3)).astype(np.float32))
    out = _C.neus_sdf_points(d_s, wb_s, pts=pts.cuda())
    with torch.no_grad():
        ref = og.sdf_only(p_sdf, cfg, pts)[:, 0]
    np.testing.assert_allclose(out.cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize('name', CFGS)
def test_networks_vs_reference_golden(name, golden_dir):
    """Same 96 points the real reference evaluated (geo_*.npz: net_sdf_out / net_sdf_grad / net_color)."""
    from vqnerf_release_amd import _C
    cfg = _cfg(name)
    g = dict(np.load(os.path.join(golden_dir, f'geo_{name}.npz')))
    _, _, wb_s, d_s, wb_c, d_c = _packed(cfg)
    rng = np.random.default_rng(3)
    pts = rng.uniform(-1.2, 1.2, (96, 3)).astype(np.float32)
    dirs = rng.normal(size=(96, 3)).astype(np.float32)
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    sdf, grad, rgb = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, pts=torch.tensor(pts).cuda(), dirs=torch.tensor(dirs).cuda())
    np.testing.assert_allclose(sdf.cpu().numpy(), g['net_sdf_out'][:, 0], rtol=0, atol=2e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), g['net_sdf_grad'], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(rgb.cpu().numpy(), g['net_color'], rtol=0, atol=2e-4)
    # gradient-only form (SDFNetwork.gradient)
    zero_col = np.zeros_like(d_c)
    sdf2, grad2, _ = _C.neus_fine_points(d_s, wb_s, zero_col, wb_c, pts=torch.tensor(pts).cuda(), dirs=torch.tensor(dirs).cuda())
    assert torch.equal(grad2, grad) and torch.equal(sdf2, sdf)


@pytest.mark.parametrize('name', CFGS)
def test_fine_points_ray_mode_vs_oracle(name):
    from oracle import geo as og
    from vqnerf_release_amd import _C
    cfg = _cfg(name)
    p_sdf, p_col, wb_s, d_s, wb_c, d_c = _packed(cfg)
    B, S = 37, 21                                  # ragged: P = 777 is not a multiple of 32
    o, d, near, far = map(torch.tensor, og.make_rays(B, 2))
    z = near + (far - near) * torch.linspace(0, 1, S)[None, :]
    sdf, grad, rgb = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, rays_o=o.cuda(), rays_d=d.cuda(), z=z.contiguous().cuda())
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
    # sdf-only kernel agrees with the fine kernel's sdf bit for bit (same packs, same order)
    s2 = _C.neus_sdf_points(d_s, wb_s, rays_o=o.cuda(), rays_d=d.cuda(), z=z.contiguous().cuda())
    assert torch.equal(s2, sdf)


def test_fine_points_fp64_error_budget():
    """How far are we from fp64 truth, next to the fp32 torch-CPU oracle?  (printed, loosely asserted)"""
    from oracle import geo as og
    from vqnerf_release_amd import _C
    cfg = og.FULL_CFG
    p_sdf, p_col, wb_s, d_s, wb_c, d_c = _packed(cfg)
    rng = np.random.default_rng(7)
    pts = torch.tensor(rng.uniform(-1.0, 1.0, (512, 3)).astype(np.float32))
    dirs = torch.nn.functional.normalize(torch.tensor(rng.normal(size=(512, 3)).astype(np.float32)), dim=1)
    sdf, grad, rgb = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, pts=pts.cuda(), dirs=dirs.cuda())
    p64 = {k: v.double() for k, v in p_sdf.items()}
    c64 = {k: v.double() for k, v in p_col.items()}
    with torch.no_grad():
        y64 = og.sdf_forward(p64, cfg, pts.double())
        y32 = og.sdf_forward(p_sdf, cfg, pts)
    g64 = og.sdf_gradient(p64, cfg, pts.double())
    g32 = og.sdf_gradient(p_sdf, cfg, pts)
    with torch.no_grad():
        c64o = og.color_forward(c64, cfg, pts.double(), g64, dirs.double(), y64[:, 1:])
    e_hip = (sdf.cpu().double() - y64[:, 0]).abs().max().item()
    e_cpu = (y32[:, 0].double() - y64[:, 0]).abs().max().item()
    eg_hip = (grad.cpu().double() - g64).abs().max().item()
    eg_cpu = (g32.double() - g64).abs().max().item()
    ec_hip = (rgb.cpu().double() - c64o).abs().max().item()
    print(f'max|err| vs fp64: sdf hip {e_hip:.2e} cpu32 {e_cpu:.2e}; grad hip {eg_hip:.2e} cpu32 {eg_cpu:.2e}; rgb hip {ec_hip:.2e}')
    assert e_hip < 1e-5 and eg_hip < 2e-4 and ec_hip < 1e-4


def test_narrow_layers_fewer_tiles_than_waves():
    """A layer with fewer 32-feature tiles than waves (here the 25-wide layer before the skip of a 64-wide net) leaves some
    waves idle in that layer; they must still hand the next layer its prefetched weight fragments."""
    import torch
    from vqnerf_release_amd.geo.models.fields import SDFNetwork
    torch.manual_seed(1)
    sdf = SDFNetwork(d_in=3, d_out=65, d_hidden=64, n_layers=4, skip_in=(2,), multires=6).cuda()
    with torch.no_grad():
        for p in sdf.parameters():
            p.add_(0.05 * torch.randn_like(p))
    pts = torch.tensor(np.random.default_rng(0).uniform(-1, 1, (333, 3)).astype(np.float32)).cuda()
    with torch.no_grad():
        got, ref = sdf.sdf(pts), sdf.forward(pts)[:, :1]
        g_hip = sdf.gradient(pts)
    g_ref = sdf.gradient(pts.clone().requires_grad_(True))
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(g_hip.cpu().numpy(), g_ref.detach().cpu().numpy(), rtol=0, atol=3e-4)
