"""CPU: analytic known-answer tests for the decomp oracle (oracle/decomp.py, oracle/vq_strict.c).

The decomp reference (TensorFlow/Sonnet) cannot run in the build image and ships no fixtures, so
these closed-form cases are all that pins this half of the oracle ("parity unpinned")."""
import math

import numpy as np
import pytest
import torch

from oracle import decomp as od
from oracle import vq_strict as vs


def test_vq_known_nearest_and_ties():
    C = torch.eye(4)[:, :3].contiguous()                    # D=4, K=3 : unit codes e0,e1,e2
    x = torch.tensor([[0.9, 0.1, 0.0, 0.0], [0.0, 0.2, 0.7, 0.1], [0.5, 0.5, 0.0, 0.0], [0.0, 0.0, 0.0, 1.0]])
    r = od.vq_ema_call(x, C, None, None, is_training=False)
    assert r['encoding_indices'].tolist() == [0, 2, 0, 0]   # row 2: tie 0/1 -> lowest; row 3: 3-way tie -> 0
    np.testing.assert_allclose(r['quantize'].numpy(), C.t()[[0, 2, 0, 0]].numpy())
    # commitment loss is ONE scalar over all N*D entries (vq_layers.py:302)
    want = 0.1 * ((C.t()[[0, 2, 0, 0]] - x) ** 2).mean()
    assert abs(r['loss'].item() - want.item()) < 1e-8
    # perplexity: p = [3/4, 0, 1/4]
    p = np.array([0.75, 0.0, 0.25])
    assert abs(r['perplexity'].item() - math.exp(-(p * np.log(p + 1e-10)).sum())) < 1e-6


def test_vq_code_dropout_mask():
    rng = np.random.default_rng(0)
    x = torch.tensor(rng.uniform(0, 1, (50, 8)).astype(np.float32))
    C = torch.tensor(rng.uniform(0, 1, (8, 5)).astype(np.float32))
    roll = torch.tensor([[0.9, 0.05, 0.9, 0.05, 0.05]])
    r = od.vq_ema_call(x, C, None, None, False, thres=torch.tensor(0.5), roll=roll)
    assert set(r['encoding_indices'].tolist()) <= {0, 2}     # only codes with roll >= thres survive
    full = od.vq_distances(x, C)
    assert torch.all(r['distances'][:, [1, 3, 4]] == full.max())
    # all codes dropped -> every distance equals the max -> index 0
    r = od.vq_ema_call(x, C, None, None, False, thres=torch.tensor(0.5), roll=torch.zeros(1, 5))
    assert r['encoding_indices'].tolist() == [0] * 50


def test_ema_zero_debias_first_two_updates():
    e = od.EMA(0.999, (3,))
    v1 = torch.tensor([1.0, 2.0, 3.0])
    a1 = e(v1)
    np.testing.assert_allclose(a1.numpy(), v1.numpy(), rtol=1e-4)       # average == v after the first update
    v2 = torch.tensor([3.0, 2.0, 1.0])
    a2 = e(v2)
    h2 = 0.999 * 0.001 * v1 + 0.001 * v2
    np.testing.assert_allclose(a2.numpy(), (h2 / (1 - 0.999 ** 2)).numpy(), rtol=1e-4)
    assert e.counter == 2


def test_vq_ema_update_laplace_and_unused_fallback():
    x = torch.tensor([[1.0, 0.0], [0.8, 0.2], [0.0, 1.0]])
    C = torch.tensor([[1.0, 0.0, 0.6], [0.0, 1.0, -0.8]])              # K=3, code 2 never wins
    K = 3
    r = od.vq_ema_call(x, C, od.EMA(0.999, (K,)), od.EMA(0.999, (2, K)), True)
    assert r['encoding_indices'].tolist() == [0, 0, 1]
    cs = torch.tensor([2.0, 1.0, 0.0]); n = 3.0
    cs_s = (cs + 1e-5) / (n + K * 1e-5) * n
    assert abs(cs_s.sum().item() - n) < 1e-5                            # Laplace smoothing keeps the total
    dw = torch.tensor([[1.8, 0.0, 0.0], [0.2, 1.0, 0.0]])
    want = dw / cs_s[None]
    want[:, 2] = C[:, 2]                                                 # unused -> the (normalised) codebook column
    np.testing.assert_allclose(r['update'].numpy(), want.numpy(), rtol=2e-4, atol=1e-6)


def test_strict_c_agrees_with_fp64_where_gap_is_clear():
    rng = np.random.default_rng(1)
    for K in (8, 15, 16, 64):
        x = rng.uniform(0, 1, (4000, 256)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        C = rng.uniform(0, 1, (256, K)).astype(np.float32)
        C /= np.linalg.norm(C, axis=0, keepdims=True)
        idx, dist, quant = vs.assign(x, C)
        d64 = od.vq_distances(torch.tensor(x, dtype=torch.float64), torch.tensor(C, dtype=torch.float64)).numpy()
        assert np.abs(dist - d64).max() < 1e-5
        top2 = np.sort(d64, 1)[:, :2]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-5
        assert clear.mean() > 0.98
        assert np.array_equal(idx[clear], d64.argmin(1)[clear])
        np.testing.assert_array_equal(quant, C.T[idx])


def test_strict_c_masked_and_ragged():
    rng = np.random.default_rng(2)
    x = rng.normal(size=(33, 20)).astype(np.float32)                    # D not a multiple of 16
    C = rng.normal(size=(20, 7)).astype(np.float32)
    sel = np.array([1, 0, 1, 1, 0, 0, 1], np.float32)
    idx, dist, _ = vs.assign(x, C, sel)
    full = od.vq_distances(torch.tensor(x, dtype=torch.float64), torch.tensor(C, dtype=torch.float64)).numpy()
    ref = np.where(sel[None] > 0, full, full.max())
    assert np.array_equal(idx, ref.argmin(1))
    idx0, _, _ = vs.assign(x, C, np.zeros(7, np.float32))
    assert (idx0 == 0).all()
    # empty input
    idx_e, _, _ = vs.assign(np.zeros((0, 20), np.float32), C)
    assert idx_e.shape == (0,)


def test_brdf_closed_forms():
    N, L = 5, 7
    rng = np.random.default_rng(3)
    n = torch.tensor([[0.0, 0.0, 1.0]]).repeat(N, 1)
    v = od.safe_l2_normalize(torch.tensor(rng.uniform(0.1, 1, (N, 3)).astype(np.float32)), 1)
    l = od.safe_l2_normalize(torch.tensor(rng.uniform(0.1, 1, (N, L, 3)).astype(np.float32)), 2)
    albedo = torch.tensor(rng.uniform(0, 1, (N, 3)).astype(np.float32))
    # rough = 1, f0 = 0 -> alpha = 1: D = 1/pi, G1(c) = 2c/(c+1), F = (1 - h.v)^5
    brdf, glossy, diffuse = od.get_brdf(l, v, n, albedo, torch.ones(N, 1), torch.zeros(N, 3))
    h = od.safe_l2_normalize(l + v[:, None], 2)
    hv = (h * v[:, None]).sum(-1)
    ln, vn = l[..., 2], v[:, 2][:, None]
    want = (1 - hv) ** 5 * (2 * ln / (ln + 1)) * (2 * vn / (vn + 1)) / math.pi / (4 * ln * vn)
    np.testing.assert_allclose(glossy[..., 0].numpy(), want.numpy(), rtol=2e-3, atol=1e-12)  # (1-h.v)^5 cancels in fp32
    np.testing.assert_allclose(diffuse.numpy(), (albedo / math.pi)[:, None].expand(N, L, 3).numpy(), rtol=1e-6)
    # grazing view exactly in the tangent plane: n.v = 0 -> divide_no_nan -> glossy = 0
    v0 = torch.tensor([[1.0, 0.0, 0.0]]).repeat(N, 1)
    _, g0, _ = od.get_brdf(l, v0, n, albedo, torch.full((N, 1), 0.5), torch.full((N, 3), 0.04))
    assert torch.all(g0 == 0)


def test_shading_white_light_lambertian():
    lxyz, lareas = od.gen_light_xyz(16, 32)
    lxyz, lareas = od.T(lxyz.reshape(-1, 3)), od.T(lareas.reshape(-1))
    xyz = torch.zeros(3, 3)
    n = torch.tensor([[0.0, 0.0, 1.0], [0.0, 1.0, 0.0], [0.6, 0.0, 0.8]])
    albedo = torch.tensor([[0.2, 0.4, 0.6]]).repeat(3, 1)
    l = od.calc_ldir(lxyz, xyz)
    brdf = (albedo / math.pi)[:, None].expand(3, 512, 3)
    rgb = od.render_integrate(brdf, l, n, lareas, torch.ones(16, 32, 3))
    # int_{hemisphere} cos dw = pi  -> rgb ~= albedo (quadrature of a 16x32 lat-long grid)
    np.testing.assert_allclose(rgb.numpy(), albedo.numpy(), rtol=0.04)
    # a light behind the surface contributes nothing; lvis = 0 kills everything
    rgb0 = od.render_integrate(brdf, l, n, lareas, torch.ones(16, 32, 3), lvis=torch.zeros(3, 512))
    assert torch.all(rgb0 == 0)


def test_srgb_roundtrip_and_thresholds():
    x = torch.linspace(0, 1, 1001)
    np.testing.assert_allclose(od.srgb2linear(od.linear2srgb(x)).numpy(), x.numpy(), atol=2e-6)
    assert abs(od.linear2srgb(torch.tensor(0.0031308)).item() - 0.0031308 * 12.92) < 1e-7
    assert od.linear2srgb(torch.tensor(1.7)).item() == pytest.approx(1.0)   # clips first (img.py:155)


def test_mlp_skip_concat_order():
    # mlp.py:45-49: output of the skip layer is concat(y, x_input): hidden first, input last
    spec = dict(widths=[2, 1], act=[None, None], skip_at=[0], d_in=3)
    W0 = torch.tensor([[1.0, 0.0], [0.0, 1.0], [0.0, 0.0]]); b0 = torch.zeros(2)
    W1 = torch.tensor([[1.0], [10.0], [100.0], [1000.0], [10000.0]]); b1 = torch.zeros(1)
    y = od.mlp_forward([(W0, b0), (W1, b1)], spec, torch.tensor([[1.0, 2.0, 3.0]]))
    assert y.item() == 1 + 20 + 100 + 2000 + 30000
    assert od.layer_in_dims(od.net_specs()['fine_enc']) == [63, 128, 128, 191]
    assert od.layer_in_dims(od.net_specs()['diff_vq']) == [256, 256, 384]


def test_model_call_and_loss_shapes_and_grads():
    p, specs = od.make_model_params(0, K=15)
    pt = {k: ([(od.T(W).requires_grad_(), od.T(b).requires_grad_()) for W, b in v] if isinstance(v, list)
              else od.T(v).requires_grad_()) for k, v in p.items()}
    b = {k: od.T(v) for k, v in od.make_points(64).items()}
    lxyz, la = od.gen_light_xyz(16, 32)
    out = od.model_call(pt, specs, b, od.T(lxyz.reshape(-1, 3)), od.T(la.reshape(-1)),
                        od.EMA(0.999, (15,)), od.EMA(0.999, (256, 15)))
    loss, ld = od.compute_loss(out, b['rgb'], pt['codebook_raw'])
    assert loss.shape == (64,)
    (loss.sum() / 1024).backward()
    assert all(torch.isfinite(W.grad).all() for W, _ in pt['fine_enc'])
    assert torch.isfinite(pt['codebook_raw'].grad).all() and pt['codebook_raw'].grad.abs().sum() > 0
    assert out['vq']['update'].shape == (256, 15)


def test_gamma_curve_and_display_transfer_of_non_nerf_data():
    """data types 'dtu' / 'hw' (vq_nfr.py:715-716, :736-745): `(rgb * bias) ** index` with the index clipped to [0, 5] (identity
    gradient), no sRGB transfer on `pred` (:638, :676)."""
    g = od.gamma_param(torch.tensor([1.3]), torch.tensor([7.0], requires_grad=True))
    assert g.tolist() == [pytest.approx(1.3), 5.0]
    gi = torch.tensor([7.0], requires_grad=True)
    od.gamma_param(torch.tensor([1.3]), gi)[1].backward()
    assert gi.grad.item() == 1.0                                        # clip_by_value_preserve_gradient
    # a Lambertian point under a white sky: sum = albedo * S (S ~ 1) -> curve -> clip
    lxyz, lareas = od.gen_light_xyz(16, 32)
    xyz, n, o = torch.zeros(1, 3), torch.tensor([[0.0, 0.0, 1.0]]), torch.tensor([[0.0, 0.0, 4.0]])
    l = od.calc_ldir(od.T(lxyz), xyz)
    brdf = (torch.tensor([[0.6, 0.3, 0.1]]) / od.PI)[:, None, :].expand(1, 512, 3)
    plain = od.render_integrate(brdf, l, n, od.T(lareas), torch.ones(16, 32, 3))
    same = od.render_integrate(brdf, l, n, od.T(lareas), torch.ones(16, 32, 3), gamma=od.gamma_param(torch.ones(1), torch.ones(1)))
    np.testing.assert_allclose(same.numpy(), plain.numpy(), rtol=1e-6)
    curved = od.render_integrate(brdf, l, n, od.T(lareas), torch.ones(16, 32, 3), gamma=torch.tensor([1.3, 0.8]))
    np.testing.assert_allclose(curved.numpy(), ((plain * 1.3) ** 0.8).numpy(), rtol=1e-6)
    big = od.render_integrate(brdf, l, n, od.T(lareas), 50 * torch.ones(16, 32, 3), gamma=torch.tensor([1.3, 0.8]))
    assert big.max().item() == 1.0                                      # tonemapping clip AFTER the curve
    assert torch.equal(od.displayed(plain, 'dtu'), plain) and torch.equal(od.displayed(plain, 'nerf'), od.linear2srgb(plain))


@pytest.mark.parametrize('data_type', ['nerf', 'hw'])
def test_fast_render_probe_loop_is_dst_env_per_probe(data_type):
    p, specs = od.make_model_params(seed=0, K=8)
    pt = {k: ([(od.T(W), od.T(b)) for W, b in v] if isinstance(v, list) else od.T(v)) for k, v in p.items()}
    lxyz, lareas = od.gen_light_xyz(16, 32)
    pts = od.make_points(40, seed=5, lvis=(data_type == 'nerf'))
    ob = {k: od.T(v) for k, v in pts.items()}
    rng = np.random.default_rng(0)
    probes = [od.T(rng.uniform(0, 2, (16, 32, 3)).astype(np.float32)) for _ in range(3)]
    gamma = None if data_type == 'nerf' else torch.tensor([1.3, 0.8])
    fr = od.fast_render(pt, specs, ob, od.T(lxyz), od.T(lareas), data_type=data_type, gamma=gamma, probes=probes, dst_env=probes[2])
    assert fr['rgb_probes'].shape == (40, 3, 3)
    np.testing.assert_allclose(fr['rgb_probes'][:, 2].numpy(), fr['rgb'].numpy(), rtol=0, atol=1e-6)
    # the model light as a "probe" reproduces call()'s main-branch render
    light = pt['light'].clamp(min=0)
    fr2 = od.fast_render(pt, specs, ob, od.T(lxyz), od.T(lareas), data_type=data_type, gamma=gamma, probes=[light])
    mc = od.model_call(pt, specs, ob, od.T(lxyz), od.T(lareas), None, None, mode='vali', data_type=data_type, gamma=gamma)
    np.testing.assert_allclose(fr2['rgb_probes'][:, 0].numpy(), od.displayed(mc['rgb'], data_type).numpy(), rtol=0, atol=1e-6)
    assert 'rgb' not in fr2


def _toy(K=8, n=60, data_type='nerf'):
    p, specs = od.make_model_params(seed=0, K=K)
    W, b = p['bottleneck'][-1]
    p['bottleneck'][-1] = (W * 25.0, b)
    pt = {k: ([(od.T(W), od.T(b)) for W, b in v] if isinstance(v, list) else od.T(v)) for k, v in p.items()}
    zc = od.pred_enc(pt, specs, od.T(od.make_points(K, seed=77)['xyz']))
    pt['codebook_raw'] = zc.t().contiguous()
    lxyz, lareas = od.gen_light_xyz(16, 32)
    ob = {k: od.T(v) for k, v in od.make_points(n, seed=5, lvis=(data_type == 'nerf')).items()}
    return pt, specs, ob, od.T(lxyz), od.T(lareas)


def test_remaining_entry_points_known_answers():
    """Known answers for the statements of vq_nfr.py:183-532 added in round 3 (the reference holds no fixtures for them)."""
    K = 8
    pt, specs, ob, lxyz, lareas = _toy(K)
    # init_z feeds init_mat; init_mat = [albedo | spec | rough] of the main heads, albedo + spec = basecolor
    z = od.init_z(pt, specs, ob)
    mat = od.init_mat(pt, specs, z)
    base, ks, rough = od.heads(pt, specs, z, vq=False)
    assert mat.shape == (60, 7)
    np.testing.assert_allclose((mat[:, :3] + mat[:, 3:6]).numpy(), base.numpy(), atol=1e-7)
    np.testing.assert_array_equal(mat[:, 6:].numpy(), rough.numpy())
    # a point's own encoder output as a code: that point is assigned to it, distance ~ 0
    pt['codebook_raw'] = z[:K].t().contiguous()
    emb = od.fast_embed(pt, specs, ob)['embed']
    assert emb[:K].tolist() == list(range(1, K + 1))
    # dropping a code (threshold 1) hands its rows to other codes; kept codes keep theirs
    thres = np.array([0, 0, 0, 0, 1, 1, 1, 1], np.float32)
    emb_d = od.fast_embed(pt, specs, ob, thres=thres)['embed']
    assert emb_d.max() <= 4 and torch.equal(emb_d[emb <= 4], emb[emb <= 4])
    # vis_mat: same indices, CONTINUOUS-branch materials (not the VQ heads)
    vm = od.vis_mat(pt, specs, ob, thres=thres)
    assert torch.equal(vm['embed'], emb_d)
    np.testing.assert_array_equal(vm['albedo'].numpy(), mat[:, :3].numpy())
    # vq_test: usage marks exactly the codes that won rows; rgb IS vqrgb; with every row on its own code the commitment term is ~0
    vt = od.vq_test(pt, specs, ob, lxyz, lareas, thres=thres)
    assert vt['usage'].shape == (1, K) and vt['usage'][0, 4:].sum() == 0
    assert set((vt['usage'][0].nonzero()[:, 0] + 1).tolist()) == set(emb_d.unique().tolist())
    assert vt['rgb'] is vt['vq_rgb']
    pt['codebook_raw'] = z[:K].t().contiguous()
    first = {k: v[:K] for k, v in ob.items()}
    assert float(od.vq_test(pt, specs, first, lxyz, lareas)['vqloss']) < 1e-12
    # thresholds strictly inside (0, 1) need the draw
    with pytest.raises(AssertionError):
        od.fast_embed(pt, specs, ob, thres=np.full(K, 0.5, np.float32))


def test_fast_render_edit_and_olat_statement():
    pt, specs, ob, lxyz, lareas = _toy(8)
    base = od.fast_render(pt, specs, ob, lxyz, lareas)
    em = torch.zeros(60, 3)
    em[:20, 0] = 1.0
    em[:, 2] = 1.0                                                        # only channel 0 counts (:290)
    ed = od.fast_render(pt, specs, ob, lxyz, lareas, edit_mask=em,
                        edit_material={'diff': [0.5, 0.4, 0.3], 'spec': [-1, 0, 0], 'rough': [0.9]})
    assert torch.equal(ed['albedo'][:20], torch.tensor([[0.5, 0.4, 0.3]]).expand(20, 3))
    assert torch.equal(ed['albedo'][20:], base['albedo'][20:]) and torch.equal(ed['spec'], base['spec'])
    assert torch.equal(ed['rough'][:20], torch.full((20, 1), 0.9)) and torch.equal(ed['basecolor'], base['basecolor'])
    # the reference never renders OLAT maps (vq_nfr.py:733): flag without maps -> no key
    assert 'rgb_olat' not in od.fast_render(pt, specs, ob, lxyz, lareas, relight_olat=True)
    olat = od.novel_olat(olat_inten=200.0, ambient_inten=0.5, white_bg=True)
    assert list(olat) == ['0004-0000', '0004-0008', '0004-0016', '0004-0024']
    m = olat['0004-0008']
    assert m.shape == (16, 32, 3) and float(m[4, 8, 0]) == 200.5 and float(m[0, 0, 0]) == 0.5 and float(m.sum()) == 3 * (200 + 0.5 * 512)
    assert float(od.novel_olat(ambient_inten=0.5, white_bg=False)['0004-0000'][0, 0, 0]) == 0.0
    # an OLAT map rendered as `olat_maps` equals the same map rendered as a probe
    r1 = od.fast_render(pt, specs, ob, lxyz, lareas, relight_olat=True, olat_maps=[m])
    r2 = od.fast_render(pt, specs, ob, lxyz, lareas, probes=[m])
    assert torch.equal(r1['rgb_olat'], r2['rgb_probes'])
    # gen_embed adds the indices and nothing else changes
    ge = od.fast_render(pt, specs, ob, lxyz, lareas, gen_embed=True)
    assert torch.equal(ge['embed'], od.fast_embed(pt, specs, ob)['embed']) and torch.equal(ge['albedo'], base['albedo'])
