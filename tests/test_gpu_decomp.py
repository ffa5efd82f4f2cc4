"""GPU parity of the reflectance path (decomp half) against oracle/decomp.py on the same seeded inputs.

The decomp oracle is a from-source restatement (PARITY UNPINNED against the TF reference); these tests
pin the HIP path to it.  Tolerances (fp32): MLP outputs 3e-6 abs (sigmoid outputs in (0,1), K <= 384 fmaf chains vs
torch's blocked GEMM); shaded rgb judged against the fp64 oracle (see _assert_as_accurate_as_fp32_oracle); VQ indices exact."""
import numpy as np
import pytest
import torch

from tests.decomp_util import make_config, load_oracle_params, make_batch
from tests.gpu_util import launches
from vqnerf_release_amd import _C

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.fixture(scope='module')
def setup():
    from oracle import decomp as od
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, specs = od.make_model_params(seed=0, K=15)
    model = load_oracle_params(get_model_class('vq_nfr')(make_config()), p, 'cuda')
    pt = {k: ([(od.T(W), od.T(b)) for W, b in v] if isinstance(v, list) else od.T(v)) for k, v in p.items()}
    lxyz, lareas = od.gen_light_xyz(16, 32)
    return dict(od=od, p=p, pt=pt, specs=specs, model=model, lxyz=od.T(lxyz), lareas=od.T(lareas))


def test_encoder_and_heads_vs_oracle(setup):
    od, model, pt, specs = setup['od'], setup['model'], setup['pt'], setup['specs']
    pts = od.make_points(1000, seed=1)
    xyz = torch.tensor(pts['xyz']).cuda()
    with torch.no_grad():
        z = model._pred_enc_at(xyz)
    z_ref = od.pred_enc(pt, specs, od.T(pts['xyz']))
    np.testing.assert_allclose(_np(z), z_ref.numpy(), rtol=0, atol=3e-6)
    for vq in (False, True):
        with torch.no_grad():
            got = model._all_heads(z, 'vq' if vq else 'main')
            single = (model._pred_diff_at(z, vq=vq), model._pred_spec_at(z, vq=vq), model._pred_rough_at(z, vq=vq))
        want = od.heads(pt, specs, od.T(_np(z)), vq)
        for g, s, w in zip(got, single, want):
            assert g.shape == w.shape
            np.testing.assert_allclose(_np(g), w.numpy(), rtol=0, atol=3e-6)
            assert torch.equal(g, s)                      # 3-heads-per-launch == one head per launch, bit for bit


@pytest.mark.parametrize('n', [1, 31, 32, 33, 257])
def test_chain_ragged_sizes(setup, n):
    od, model, pt, specs = setup['od'], setup['model'], setup['pt'], setup['specs']
    pts = od.make_points(n, seed=n)
    with torch.no_grad():
        z = model._pred_enc_at(torch.tensor(pts['xyz']).cuda())
    np.testing.assert_allclose(_np(z), od.pred_enc(pt, specs, od.T(pts['xyz'])).numpy(), rtol=0, atol=3e-6)


def test_encoder_and_main_heads_in_one_program_are_bit_identical(setup):
    """`enc_and_heads` (inference path: encoder + the three continuous-branch heads as ONE layer program, z never read back from
    HBM) against the two separate programs: z and every head output bit for bit."""
    od, model = setup['od'], setup['model']
    from tests.gpu_util import launches
    xyz = torch.tensor(od.make_points(20011, seed=78)['xyz']).cuda()
    with torch.no_grad():
        with launches() as rec:
            z, d, s_, r = model.enc_and_heads(xyz, 'main')
        assert rec.counts.get('vqn_mlp_chain_fwd') == 1
        z2 = model._pred_enc_at(xyz)
        d2, s2, r2 = model._all_heads(z2, 'main')
    for a, b in ((z, z2), (d, d2), (s_, s2), (r, r2)):
        assert a.shape == b.shape and torch.equal(a, b)


def test_chain_outputs_do_not_depend_on_the_launch_size(setup):
    """A point's encoder / head outputs are the same bits whether it is evaluated in a 5 k-point or a 100 k-point launch (one tile
    per persistent workgroup pass vs many; ragged last tile): the arithmetic per point has one defined order."""
    od, model = setup['od'], setup['model']
    from tests.gpu_util import launches
    n_big, n_small = 100003, 4999                                        # 3126 tiles (two-image, ragged) / 157 tiles (one-image)
    xyz = torch.tensor(od.make_points(n_big, seed=77)['xyz']).cuda()
    with torch.no_grad():
        z_big = model._pred_enc_at(xyz)
        z_small = model._pred_enc_at(xyz[:n_small].contiguous())
        assert torch.equal(z_big[:n_small], z_small)
        for fam in ('main', 'vq'):
            big = model._all_heads(z_big, fam)
            small = model._all_heads(z_big[:n_small].contiguous(), fam)
            for b, s_ in zip(big, small):
                assert torch.equal(b[:n_small], s_)
        assert torch.isfinite(z_big).all() and float(z_big[-1].abs().sum()) > 0


@pytest.mark.parametrize('n', [1, 33, 1000])
def test_split_precision_chain_vs_fp64_oracle(setup, n):
    """matrix_mode = 'f16s' (vqn_mlp_chain_fwd_f16s: f16 hi/lo operands, three f16 MFMAs per product, f32 accumulate) is an
    opt-in mode with a STATED tolerance instead of bitwise agreement: against the oracle in float64 its error must stay
    within 4x the f32 kernel's own error + 2e-6 of the output scale (measured: about 1-2x)."""
    od, model, p, specs = setup['od'], setup['model'], setup['p'], setup['specs']
    pts = od.make_points(n, seed=40 + n)
    xyz = torch.tensor(pts['xyz']).cuda()
    p64 = {k: [(torch.tensor(W, dtype=torch.float64), torch.tensor(b, dtype=torch.float64)) for W, b in v]
           for k, v in p.items() if isinstance(v, list)}
    z64 = od.pred_enc(p64, specs, torch.tensor(pts['xyz'], dtype=torch.float64))
    try:
        with torch.no_grad():
            z32 = model._pred_enc_at(xyz)
            h32 = model._all_heads(z32, 'main') + model._all_heads(z32, 'vq')
            model.matrix_mode = 'f16s'
            z16 = model._pred_enc_at(xyz)
            h16 = model._all_heads(z32, 'main') + model._all_heads(z32, 'vq')
            one = model._pred_rough_at(z32, vq=True)
    finally:
        model.matrix_mode = 'f32'
    scale = float(z64.abs().max())
    e32, e16 = np.abs(_np(z32) - z64.numpy()).max(), np.abs(_np(z16) - z64.numpy()).max()
    assert e16 <= 4 * e32 + 2e-6 * scale, (e16, e32, scale)
    assert not torch.equal(z16, z32)                                  # it really is the other kernel
    want = od.heads(p64, specs, torch.tensor(_np(z32), dtype=torch.float64), False) + \
        od.heads(p64, specs, torch.tensor(_np(z32), dtype=torch.float64), True)
    for g32, g16, w in zip(h32, h16, want):
        assert g16.shape == w.shape
        a32, a16 = np.abs(_np(g32) - w.numpy()).max(), np.abs(_np(g16) - w.numpy()).max()
        assert a16 <= 4 * a32 + 2e-6, (a16, a32)
    assert torch.equal(one, h16[5])                                   # one head per launch == three per launch, here too


@pytest.mark.gpu
@pytest.mark.parametrize('n', [1, 33, 2048, 5000])
def test_exact_split_chain_vs_fp64_oracle(setup, n):
    """matrix_mode = 'x3' (round 4): inference on the exact-split stack kernel of the trainers (vqn_refl_train_fwd_x3 with nothing kept
    for a backward: bf16 piece triples, six bf16 MFMAs per product, f32 accumulate).  Its products are f32-exact up to the dropped
    piece-pair terms (2^-24 relative), so against the oracle in float64 it must be as accurate as the f32 kernel: within 2x the f32
    kernel's own error + 1e-6 of the output scale.  Encoder + three heads of a family in ONE launch, heads alone in ONE launch."""
    od, model, p, specs = setup['od'], setup['model'], setup['p'], setup['specs']
    pts = od.make_points(n, seed=60 + n)
    xyz = torch.tensor(pts['xyz']).cuda()
    p64 = {k: [(torch.tensor(W, dtype=torch.float64), torch.tensor(b, dtype=torch.float64)) for W, b in v]
           for k, v in p.items() if isinstance(v, list)}
    z64 = od.pred_enc(p64, specs, torch.tensor(pts['xyz'], dtype=torch.float64))
    try:
        with torch.no_grad():
            z32 = model._pred_enc_at(xyz)
            h32 = model._all_heads(z32, 'main') + model._all_heads(z32, 'vq')
            model.matrix_mode = 'x3'
            _C.KernelClock.reset(True)
            zx = model._pred_enc_at(xyz)
            hx = model._all_heads(z32, 'main') + model._all_heads(z32, 'vq')
            one = model._pred_rough_at(z32, vq=True)
            zf, a, s_, r = model.enc_and_heads(xyz, 'main')
            torch.cuda.synchronize()
            clk = _C.KernelClock.summary()
    finally:
        model.matrix_mode = 'f32'
        _C.KernelClock.reset(False)
    assert clk['vqn_refl_train_fwd_x3'][0] == 5 and 'vqn_mlp_chain_fwd' not in clk, clk      # enc | main heads | vq heads | one head | enc + heads
    scale = float(z64.abs().max())
    e32, ex = np.abs(_np(z32) - z64.numpy()).max(), np.abs(_np(zx) - z64.numpy()).max()
    assert ex <= 2 * e32 + 1e-6 * scale, (ex, e32, scale)
    assert zx.shape == z32.shape and not torch.equal(zx, z32)          # it really is the other kernel
    assert torch.equal(zf, zx)                                         # the encoder rows do not depend on which heads ride along
    z_at = torch.tensor(_np(z32), dtype=torch.float64)
    want = od.heads(p64, specs, z_at, False) + od.heads(p64, specs, z_at, True)
    for g32, gx, w in zip(h32, hx, want):
        assert gx.shape == w.shape
        a32, ax = np.abs(_np(g32) - w.numpy()).max(), np.abs(_np(gx) - w.numpy()).max()
        assert ax <= 2 * a32 + 1e-6, (ax, a32)
    assert np.abs(_np(one) - _np(hx[5])).max() <= 1e-6                # one head per launch vs three per launch: the same sums
    # heads at the x3 encoder's own rows: enc + heads in one launch == the two launches
    with torch.no_grad():
        model.matrix_mode = 'x3'
        try:
            h_at = model._all_heads(zx, 'main')
        finally:
            model.matrix_mode = 'f32'
    for got, sep in zip((a, s_, r), h_at):
        assert np.abs(_np(got) - _np(sep)).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['test', 'vali'])
def test_exact_split_model_call_vs_f32(setup, mode):
    """model.call with matrix_mode = 'x3' end to end (encoder + main heads, quantiser, VQ heads, shading): every output within f32
    rounding of the f32 path, the same codes except at near-ties."""
    od, model = setup['od'], setup['model']
    batch = make_batch(od.make_points(3000, seed=77), 'cuda', bg_every=7)
    with torch.no_grad():
        ref = model.call(batch, mode=mode)[0]
        model.matrix_mode = 'x3'
        try:
            _C.KernelClock.reset(True)
            got = model.call(batch, mode=mode)[0]
            torch.cuda.synchronize()
            clk = _C.KernelClock.summary()
        finally:
            model.matrix_mode = 'f32'
            _C.KernelClock.reset(False)
    assert clk['vqn_refl_train_fwd_x3'][0] == 2 and 'vqn_mlp_chain_fwd' not in clk and 'vqn_mlp_chain_vq_fwd' not in clk, clk
    same = (got['embed'] == ref['embed']).float().mean()
    assert float(same) >= 0.995
    ok = (got['embed'] == ref['embed']).reshape(-1)
    checked = 0
    for k in ('rgb', 'albedo', 'rough', 'spec', 'vq_rgb', 'vq_albedo'):
        if k in ref and torch.is_tensor(ref[k]) and ref[k].shape[0] == ok.shape[0]:
            d = (got[k] - ref[k]).abs()[ok]
            assert float(d.max()) <= 5e-5, (k, float(d.max()))
            checked += 1
    assert checked >= 4


def _oracle_shade(od, setup, pts, mats, light, with_lvis, dtype, gamma=None):
    T = lambda a: od.T(a, dtype)
    xyz, normal, rayo = T(pts['xyz']), T(pts['normal']), T(pts['rayo'])
    lvis = T(pts['lvis']) if with_lvis else None
    lxyz, lareas = setup['lxyz'].to(dtype), setup['lareas'].to(dtype)
    surf2l = od.calc_ldir(lxyz, xyz)
    surf2c = od.calc_vdir(rayo, xyz)
    n_pred = od.normal_correct(normal, surf2c)
    out = dict(normal=n_pred, rgb=[])
    for i, (a, s, r) in enumerate(mats):
        brdf, bs, bd = od.get_brdf(surf2l, surf2c, n_pred, T(a), T(r), T(s))
        out['rgb'].append(od.render_integrate(brdf, surf2l, n_pred, lareas, T(light), lvis, gamma))
        if i == 0:
            out['rgb_diff'] = od.render_integrate(bd, surf2l, n_pred, lareas, T(light), lvis, gamma)
            out['rgb_spec'] = od.render_integrate(bs, surf2l, n_pred, lareas, T(light), lvis, gamma)
    return out


def _assert_as_accurate_as_fp32_oracle(got, ref32, ref64, what):
    """GGX with small roughness is ill-conditioned in fp32 (t = cos_m^2 (a^2 - 1) + 1 cancels to ~a^2 = rough^4), so
    two correct fp32 evaluations differ by ~1e-4 on glossy peaks.  Ground truth = the oracle in fp64; the kernel must
    be as close to it as the fp32 oracle is (max error within x3, rms within x2, median <= 2e-6)."""
    e_hip = np.abs(got.astype(np.float64) - ref64)
    e_o32 = np.abs(ref32.astype(np.float64) - ref64)
    assert e_hip.max() <= max(3.0 * e_o32.max(), 2e-5), (what, e_hip.max(), e_o32.max())
    assert np.median(e_hip) <= 2e-6, (what, np.median(e_hip))
    rms = lambda e: float(np.sqrt((e ** 2).mean()))
    assert rms(e_hip) <= max(2.0 * rms(e_o32), 2e-6), (what, rms(e_hip), rms(e_o32))


@pytest.mark.parametrize('with_lvis', [True, False])
def test_shade_kernel_vs_oracle(setup, with_lvis):
    od = setup['od']
    from vqnerf_release_amd import _C
    N = 500
    pts = od.make_points(N, seed=7)
    rng = np.random.default_rng(8)
    mats = [(rng.uniform(0, 1, (N, 3)), rng.uniform(0, 1, (N, 3)), rng.uniform(0.02, 1, (N, 1))) for _ in range(2)]
    mats[0][2][:5] = 0.0                                    # rough = 0 -> D = divide_no_nan(0, .) paths
    light = rng.uniform(0, 1, (16, 32, 3))
    r32 = _oracle_shade(od, setup, pts, mats, light, with_lvis, torch.float32)
    r64 = _oracle_shade(od, setup, pts, mats, light, with_lvis, torch.float64)
    c = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32).cuda().contiguous()
    geo = (c(pts['xyz']), c(pts['normal']), c(pts['rayo']))
    lights = (c(setup['lxyz'].reshape(-1, 3)), c(setup['lareas'].reshape(-1)), c(light.reshape(-1, 3)))
    got = _C.brdf_shade_fwd(*geo, c(pts['lvis']) if with_lvis else None, *lights,
                            [(c(a), c(s), c(r)) for a, s, r in mats], want_normal=True, want_split=True)
    np.testing.assert_array_equal(_np(got['normal']), r32['normal'].numpy())
    for i in range(2):
        _assert_as_accurate_as_fp32_oracle(_np(got['rgb'][i]), r32['rgb'][i].numpy(), r64['rgb'][i].numpy(), f'rgb{i}')
    np.testing.assert_allclose(_np(got['rgb_diff']), r32['rgb_diff'].numpy(), rtol=0, atol=2e-6)   # Lambert part: well conditioned
    _assert_as_accurate_as_fp32_oracle(_np(got['rgb_spec']), r32['rgb_spec'].numpy(), r64['rgb_spec'].numpy(), 'rgb_spec')
    # non-nerf data: learnable gamma (vq_nfr.py:715-716)
    gam = torch.tensor([1.3, 0.8]).cuda()
    got_g = _C.brdf_shade_fwd(*geo, None, *lights, [(c(mats[0][0]), c(mats[0][1]), c(mats[0][2]))], gamma=gam)
    g32 = _oracle_shade(od, setup, pts, mats[:1], light, False, torch.float32, gamma=(1.3, 0.8))
    g64 = _oracle_shade(od, setup, pts, mats[:1], light, False, torch.float64, gamma=(1.3, 0.8))
    _assert_as_accurate_as_fp32_oracle(_np(got_g['rgb'][0]), g32['rgb'][0].numpy(), g64['rgb'][0].numpy(), 'gamma')


def test_shade_clips_with_identity_gradient_inside_the_kernel(setup):
    """raw = 2 (round 4: the training path of data_type 'nerf'): the plain sums through tfp's clip_by_value_preserve_gradient inside the
    shading kernel == raw = 1 followed by vqn_clip_preserve, bit for bit (bright lights so that the clip bites)."""
    od = setup['od']
    N = 700
    pts = od.make_points(N, seed=23)
    rng = np.random.default_rng(24)
    c = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32).cuda().contiguous()
    mats = [(c(rng.uniform(0, 1, (N, 3))), c(rng.uniform(0, 1, (N, 3))), c(rng.uniform(0.02, 1, (N, 1)))) for _ in range(2)]
    geo = (c(pts['xyz']), c(pts['normal']), c(pts['rayo']))
    lights = (c(setup['lxyz'].reshape(-1, 3)), c(setup['lareas'].reshape(-1)), c(rng.uniform(0, 6, (512, 3))))
    a = _C.brdf_shade_fwd(*geo, c(pts['lvis']), *lights, mats, raw=1)
    b = _C.brdf_shade_fwd(*geo, c(pts['lvis']), *lights, mats, raw=2)
    for s in range(2):
        want = _C.clip_preserve(a['rgb'][s], 0.0, 1.0)
        assert torch.equal(b['rgb'][s], want)
        assert float((a['rgb'][s] > 1).float().mean()) > 0.01        # (the clip did something)


def test_shade_reads_visibility_rows_in_place(setup):
    """vqn_brdf_shade_fwd_rows: the foreground gather of the visibility buffer (vq_nfr.py:558-559) folded into the kernel gives
    bit for bit what the gathered copy gives; the models use it on the inference path (LazyRows)."""
    od = setup['od']
    from vqnerf_release_amd import _C
    n_view, N = 900, 500
    pts = od.make_points(n_view, seed=17)
    rng = np.random.default_rng(18)
    rows = np.sort(rng.choice(n_view, N, replace=False)).astype(np.int64)
    c = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32).cuda().contiguous()
    mats = [(c(rng.uniform(0, 1, (N, 3))), c(rng.uniform(0, 1, (N, 3))), c(rng.uniform(0.02, 1, (N, 1)))) for _ in range(2)]
    geo = (c(pts['xyz'][rows]), c(pts['normal'][rows]), c(pts['rayo'][rows]))
    lights = (c(setup['lxyz'].reshape(-1, 3)), c(setup['lareas'].reshape(-1)), c(rng.uniform(0, 1, (512, 3))))
    full = c(pts['lvis'])
    a = _C.brdf_shade_fwd(*geo, full[torch.tensor(rows).cuda()].contiguous(), *lights, mats, want_split=True)
    b = _C.brdf_shade_fwd(*geo, full, *lights, mats, want_split=True, lvis_rows=torch.tensor(rows).cuda())
    for k in ('rgb_diff', 'rgb_spec', 'normal'):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a['rgb'][0], b['rgb'][0]) and torch.equal(a['rgb'][1], b['rgb'][1])
    from vqnerf_release_amd.decomp.nerfactor.models.nfr_unit import LazyRows
    lz = LazyRows(full, torch.tensor(rows).cuda())
    assert lz.shape == (N, 512) and torch.equal(lz.dense(), full[torch.tensor(rows).cuda()])


def test_shade_known_answers():
    """Analytic pins: a Lambertian point under a white unit sky integrates to ~albedo; a light
    behind the surface contributes nothing; lvis = 0 everywhere gives black."""
    from oracle import decomp as od
    from vqnerf_release_amd import _C
    lxyz, lareas = od.gen_light_xyz(16, 32)
    c = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32).cuda().contiguous()
    xyz = c([[0, 0, 0.0]]); normal = c([[0, 0, 1.0]]); rayo = c([[0, 0, 4.0]])
    alb = c([[0.6, 0.3, 0.1]]); spec = c([[0.0, 0.0, 0.0]]); rough = c([[1.0]])
    light = c(np.ones((512, 3)))
    out = _C.brdf_shade_fwd(xyz, normal, rayo, None, c(lxyz.reshape(-1, 3)), c(lareas.reshape(-1)), light, [(alb, spec, rough)],
                            want_split=True)
    # diffuse part: albedo/pi * sum_front cos*area  ~= albedo (discretised hemisphere integral of cos = pi)
    np.testing.assert_allclose(_np(out['rgb_diff'])[0], [0.6, 0.3, 0.1], rtol=0.02)
    black = _C.brdf_shade_fwd(xyz, normal, rayo, c(np.zeros((1, 512))), c(lxyz.reshape(-1, 3)), c(lareas.reshape(-1)), light,
                              [(alb, spec, rough)])
    assert float(black['rgb'][0].abs().max()) == 0.0
    flipped = _C.brdf_shade_fwd(xyz, c([[0, 0, -1.0]]), rayo, None, c(lxyz.reshape(-1, 3)), c(lareas.reshape(-1)), light,
                                [(alb, spec, rough)])
    assert torch.equal(flipped['normal'], normal)           # camera-facing correction
    assert torch.equal(flipped['rgb'][0], out['rgb'][0])


@pytest.mark.parametrize('mode', ['vali', 'train'])
def test_model_call_vs_oracle(setup, mode):
    od, p, specs = setup['od'], setup['p'], setup['specs']
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    model = load_oracle_params(get_model_class('vq_nfr')(make_config()), p, 'cuda')
    N = 600
    pts = od.make_points(N, seed=3)
    batch = make_batch(pts, 'cuda', bg_every=7)
    keep = np.ones(N, bool); keep[::7] = False
    pt = setup['pt']
    ob = {k: od.T(v[keep]) for k, v in pts.items()}
    ema_cs, ema_dw = od.EMA(0.999, (15,)), od.EMA(0.999, (256, 15))
    want = od.model_call(pt, specs, ob, setup['lxyz'], setup['lareas'], ema_cs, ema_dw, mode=mode)
    with torch.no_grad():
        pred, gt, lk, to_vis = model.call(batch, mode=mode)
    m = torch.tensor(keep).cuda()
    np.testing.assert_array_equal(_np(pred['rgb'][~m]), 0.0)                 # background rays stay zero
    np.testing.assert_allclose(_np(pred['rgb'][m]), od.linear2srgb(want['rgb']).numpy(), rtol=0, atol=1e-4)
    np.testing.assert_allclose(_np(lk['rgb']), want['rgb'].numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(_np(lk['vqrgb']), want['vq_rgb'].numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(_np(pred['albedo'][m]), want['albedo'].numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(_np(pred['rough'][m]), want['rough'].numpy(), rtol=0, atol=5e-6)
    np.testing.assert_allclose(_np(lk['z']), want['z_vq'].numpy(), rtol=0, atol=1e-6)
    np.testing.assert_allclose(float(lk['vqloss']), float(want['vq']['loss']), rtol=1e-4)
    if mode != 'train':
        np.testing.assert_array_equal(_np(pred['embed'][m])[:, 0], want['embed'].numpy())    # VQ indices exact
        np.testing.assert_allclose(_np(pred['rgb_diff'][m]), want['rgb_diff'].numpy(), rtol=0, atol=2e-5)
        np.testing.assert_allclose(_np(pred['vq_albedo'][m]), want['vq_albedo'].numpy(), rtol=0, atol=5e-6)
    else:
        # the EMA moved the codebook (vq_nfr.py:582-583)
        np.testing.assert_allclose(_np(model._codebook), want['vq']['update'].numpy(), rtol=0, atol=2e-6)
    loss, ld = model.compute_loss(pred, gt, **dict(lk))
    # compute_loss runs after call(): in train mode it sees the codebook the EMA has just written (vq_nfr.py:582-583, :956)
    cb_for_loss = want['vq']['update'] if mode == 'train' else pt['codebook_raw']
    wl, wd = od.compute_loss(want, ob['rgb'], cb_for_loss, mode=mode)
    np.testing.assert_allclose(_np(loss), wl.numpy(), rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize('backend', ['hip', 'hip-prog', 'torch'])
def test_training_step_grads_vs_oracle(setup, backend, monkeypatch):
    """Training path of the model -- 'hip': encoder / heads forward + backward on the dedicated exact-split kernels
    (`vqn_refl_train_fwd_x3` / `_bwd_x3`, round 4), 'hip-prog': on the interpreted tile programs (`vqn_tile_program`), both with the
    batched contractions (`vqn_wgrad_partials`) and the fused shading forward / backward kernels under autograd; 'torch': torch
    statements only; all with the HIP VQ kernels -- d loss / d parameters vs the CPU oracle.  Asserts which kernels each backend launched."""
    od, p, specs = setup['od'], setup['p'], setup['specs']
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    model = load_oracle_params(get_model_class('vq_nfr')(make_config()), p, 'cuda')
    if backend == 'hip-prog':
        monkeypatch.setenv('VQN_REFL_TRAIN', 'prog')
    prog, backend = backend == 'hip-prog', backend.split('-')[0]
    model.train_backend = backend
    N = 256
    pts = od.make_points(N, seed=9)
    batch = make_batch(pts, 'cuda')
    with launches() as rec:
        pred, gt, lk, _ = model.call(batch, mode='train')
        loss, _ = model.compute_loss(pred, gt, **dict(lk))
        loss.sum().div(N).backward()
    hip = backend == 'hip'
    assert rec.ran('vqn_tile_program') == (hip and prog) and rec.ran('vqn_refl_train_bwd_x3') == (hip and not prog)
    assert rec.ran('vqn_refl_train_fwd_x3') == (hip and not prog) and rec.ran('vqn_brdf_shade_bwd') == hip and rec.ran('vqn_wgrad_partials') == hip
    assert rec.ran('vqn_vq_quantize_rows_train') == hip and rec.ran('vqn_vq_assign') == (not hip)     # (the VQ kernels run under every backend)
    # oracle with torch autograd on the CPU
    pt = {k: ([(od.T(W).requires_grad_(True), od.T(b).requires_grad_(True)) for W, b in v] if isinstance(v, list)
              else od.T(v).requires_grad_(True)) for k, v in p.items()}
    ob = {k: od.T(v) for k, v in pts.items()}
    want = od.model_call(pt, specs, ob, setup['lxyz'], setup['lareas'], od.EMA(0.999, (15,)), od.EMA(0.999, (256, 15)), mode='train')
    # compute_loss reads the codebook AFTER the EMA move of call() (vq_nfr.py:582-583, :956): its gradient -- the code-separation term
    # (:958-971) is the only path into `_codebook` -- is taken at the moved values
    cb_leaf = want['vq']['update'].detach().clone().requires_grad_(True)
    wl, _ = od.compute_loss(want, ob['rgb'], cb_leaf, mode='train')
    wl.sum().div(N).backward()
    ref = cb_leaf.grad.numpy()
    assert np.abs(ref).max() > 0 and model._codebook.grad is not None
    assert np.abs(_np(model._codebook.grad) - ref).max() <= 2e-3 * np.abs(ref).max(), '_codebook.grad vs the oracle'
    for name, net in model.net.items():
        for layer, (W, b) in zip(net.layers, pt[name]):
            for got, ref in ((layer.kernel.grad, W.grad), (layer.bias.grad, b.grad)):
                ref = ref.numpy()
                scale = max(np.abs(ref).max(), 1e-8)
                assert np.abs(_np(got) - ref).max() <= 2e-3 * scale + 1e-9, name
    ref = pt['light'].grad.numpy()
    assert np.abs(_np(model._light.grad) - ref).max() <= 2e-3 * np.abs(ref).max()


def test_full_image_properties(setup):
    """BASELINE-size point set (800x800): size-independent invariants of the fused inference path."""
    od, model = setup['od'], setup['model']
    N = 640000
    rng = np.random.default_rng(0)
    xyz = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
    xyz /= np.linalg.norm(xyz, axis=1, keepdims=True)
    normal = xyz.copy()
    pts = dict(xyz=xyz * 0.8, normal=normal, rayo=np.tile(np.array([[0, 0, 4.0]], np.float32), (N, 1)),
               rgb=rng.uniform(0, 1, (N, 3)).astype(np.float32))
    batch = make_batch(pts, 'cuda')[:9] + (torch.ones(N, 512, device='cuda'),)
    with torch.no_grad():
        pred, gt, lk, _ = model.call(batch, mode='vali')
        assert torch.isfinite(pred['rgb']).all() and (pred['rgb'] >= 0).all() and (pred['rgb'] <= 1).all()
        emb = pred['embed'][:, 0]
        assert emb.min() >= 1 and emb.max() <= 15
        # permutation equivariance over points, bit for bit
        perm = torch.randperm(N, device='cuda')
        b2 = tuple(t[perm] if torch.is_tensor(t) else t for t in batch)
        pred2, _, _, _ = model.call(b2, mode='vali')
        assert torch.equal(pred2['rgb'], pred['rgb'][perm])
        assert torch.equal(pred2['embed'], pred['embed'][perm])
        # chunking independence
        b3 = tuple(t[:100001] if torch.is_tensor(t) else t[:100001] for t in batch)
        pred3, _, _, _ = model.call(b3, mode='vali')
        assert torch.equal(pred3['vq_rgb'], pred['vq_rgb'][:100001])


@pytest.mark.parametrize('n_probes', [16, 5, 20, 30])
def test_relight_16_probes_single_pass(setup, n_probes):
    """fast_render(relight_probes=True) (test.py:254-266 -> vq_nfr.py:724-733): all probes in ONE shading pass must equal
    separate passes (to rounding: the radiance is multiplied in last instead of first) and the oracle within the shading tolerance.
    16 / 5 / 20 probes: the probe table staged in LDS (one chunk, a ragged chunk, two chunks); 30: beyond the LDS budget, read from
    global memory."""
    od, model, pt, specs = setup['od'], setup['model'], setup['pt'], setup['specs']
    rng = np.random.default_rng(2)
    model.novel_probes = {f'probe{i:02d}': torch.tensor(rng.uniform(0, 2, (16, 32, 3)).astype(np.float32)).cuda() for i in range(n_probes)}
    N = 300
    pts = od.make_points(N, seed=12)
    batch = make_batch(pts, 'cuda', bg_every=5)
    with torch.no_grad():
        pred, gt, lk, to_vis = model.fast_render(batch, mode='test', relight_probes=True)
        assert pred['rgb_probes'].shape == (N, n_probes, 3)
        keep = np.ones(N, bool); keep[::5] = False
        m = torch.tensor(keep).cuda()
        assert float(pred['rgb_probes'][~m].abs().max()) == 0.0
        # one probe at a time through dst_env
        for i, name in [(j, list(model.novel_probes)[j]) for j in (0, 1, n_probes // 2, n_probes - 1)]:
            one, _, _, _ = model.fast_render(batch, mode='test', dst_env=name)
            np.testing.assert_allclose(_np(one['rgb'][m]), _np(pred['rgb_probes'][m][:, i]), rtol=0, atol=3e-6)
    # oracle
    T = od.T
    ob = {k: T(v[keep]) for k, v in pts.items()}
    surf2l = od.calc_ldir(setup['lxyz'], ob['xyz']); surf2c = od.calc_vdir(ob['rayo'], ob['xyz'])
    n_pred = od.normal_correct(ob['normal'], surf2c)
    z = od.pred_enc(pt, specs, ob['xyz'])
    base, ks, rough = od.heads(pt, specs, z, False)
    brdf, _, _ = od.get_brdf(surf2l, surf2c, n_pred, (1 - ks) * base, rough, ks * base)
    for i, lp in [(j, list(model.novel_probes.values())[j]) for j in (0, 2, n_probes - 1)]:
        want = od.linear2srgb(od.render_integrate(brdf, surf2l, n_pred, setup['lareas'], lp.cpu(), ob['lvis']))
        np.testing.assert_allclose(_np(pred['rgb_probes'][m][:, i]), want.numpy(), rtol=0, atol=2e-4)
    model.novel_probes = {}


def _ref_setup(data_type, seed=5):
    from oracle import decomp as od
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, specs = od.make_ref_params(seed=seed)
    m = load_oracle_params(get_model_class('ref_nfr')(make_config(model='ref_nfr', data_type=data_type)), p, 'cuda')
    gamma = None
    if data_type != 'nerf':
        m.gamma
        with torch.no_grad():
            m._gamma_bias.fill_(1.3); m._gamma_index.fill_(0.8)
        gamma = od.gamma_param(torch.tensor([1.3]), torch.tensor([0.8]))
    pt = {k: ([(od.T(W), od.T(b)) for W, b in v] if isinstance(v, list) else od.T(v)) for k, v in p.items()}
    return od, p, pt, specs, m, gamma


def _ref_batch(od, N, data_type, seed, bg_every):
    pts = od.make_points(N, seed=seed, lvis=(data_type == 'nerf'))
    pts['ref'] = np.random.default_rng(seed + 1).uniform(0, 1, (N, 3)).astype(np.float32)
    b = make_batch({k: v for k, v in pts.items() if k != 'ref'}, 'cuda', bg_every=bg_every)
    batch = b[:9] + (torch.tensor(pts['ref']).cuda(),) + b[9:]
    keep = np.ones(N, bool)
    if bg_every:
        keep[::bg_every] = False
    return pts, batch, keep, {k: od.T(v[keep]) for k, v in pts.items()}


@pytest.mark.parametrize('data_type,matrix_mode', [('nerf', 'f32'), ('hw', 'f32'), ('hw', 'f16s')])
def test_ref_nfr_call_and_fast_render_vs_oracle(setup, data_type, matrix_mode):
    """Stage-3 model (ref_nfr.py:176-300, :303-418) on the fused kernels against the oracle's statement: `call` in vali mode
    with 3 probes, `fast_render` with the albedo / spec scale (rgb from the unscaled materials, probes from the scaled ones),
    per-example loss.  'hw' + 'f16s' is BASELINE configs[4]'s model / data type / precision mode (scripts/test/relight_hw.sh)."""
    od, p, pt, specs, m, gamma = _ref_setup(data_type)
    m.matrix_mode = matrix_mode
    lxyz, lareas = setup['lxyz'], setup['lareas']
    rng = np.random.default_rng(2)
    probes = [rng.uniform(0, 2, (16, 32, 3)).astype(np.float32) for _ in range(3)]
    m.novel_probes = {f'probe{i}': torch.tensor(a).cuda() for i, a in enumerate(probes)}
    N = 300
    pts, batch, keep, ob = _ref_batch(od, N, data_type, 21, 9)
    assert len(batch) == (11 if data_type == 'nerf' else 10)              # ref_nfr.py:180-184
    mk = torch.tensor(keep).cuda()
    want = od.ref_nfr_call(pt, specs, ob, lxyz, lareas, mode='vali', data_type=data_type, gamma=gamma, probes=[od.T(a) for a in probes])
    with torch.no_grad():
        pred, gt, lk, _ = m.call(batch, mode='vali', relight_probes=True)
    t = 1 if matrix_mode == 'f32' else 3
    np.testing.assert_allclose(_np(lk['rgb']), want['rgb'].numpy(), rtol=0, atol=t * 2e-5)
    np.testing.assert_allclose(_np(pred['rgb'][mk]), want['pred_rgb'].numpy(), rtol=0, atol=t * 1e-4)
    np.testing.assert_array_equal(_np(pred['rgb'][~mk]), 0.0)
    for k in ('albedo', 'spec', 'rough', 'ks', 'basecolor'):
        np.testing.assert_allclose(_np(pred[k][mk]), want[k].numpy(), rtol=0, atol=t * 5e-6, err_msg=k)
    np.testing.assert_allclose(_np(pred['rgb_diff'][mk]), want['rgb_diff'].numpy(), rtol=0, atol=t * 2e-5)
    np.testing.assert_allclose(_np(pred['rgb_spec'][mk]), want['rgb_spec'].numpy(), rtol=0, atol=t * 1e-4)
    np.testing.assert_allclose(_np(pred['rgb_probes'][mk]), want['rgb_probes'].numpy(), rtol=0, atol=t * 2e-4)
    np.testing.assert_array_equal(_np(pred['normal'][mk]), want['normal'].numpy())
    loss = m.compute_loss(pred, gt, **dict(lk))                           # vali: the bare tensor (ref_nfr.py:606)
    np.testing.assert_allclose(_np(loss), od.ref_nfr_loss(want, ob['rgb'], data_type).numpy(), rtol=1e-3, atol=t * 2e-6)
    assert lk['env'] is None
    # fast_render
    scale = torch.tensor([[1.2, 0.9, 0.8]])
    want_f = od.ref_nfr_fast_render(pt, specs, ob, lxyz, lareas, data_type=data_type, gamma=gamma, probes=[od.T(a) for a in probes],
                                    opt_scale=scale)
    with torch.no_grad():
        pf, gf, lkf, _ = m.fast_render(batch, mode='test', relight_probes=True, opt_scale=scale.cuda())
        p0, _, _, _ = m.fast_render(batch, mode='test')
    assert set(pf) == {'rgb', 'alpha', 'rgb_probes'} and set(p0) == {'rgb', 'alpha'}
    np.testing.assert_allclose(_np(pf['rgb'][mk]), want_f['pred_rgb'].numpy(), rtol=0, atol=t * 1e-4)
    np.testing.assert_allclose(_np(p0['rgb'][mk]), want_f['pred_rgb'].numpy(), rtol=0, atol=t * 1e-4)
    np.testing.assert_allclose(_np(lkf['rgb']), want_f['rgb'].numpy(), rtol=0, atol=t * 2e-5)
    np.testing.assert_allclose(_np(pf['rgb_probes'][mk]), want_f['rgb_probes'].numpy(), rtol=0, atol=t * 2e-4)


@pytest.mark.parametrize('data_type', ['nerf', 'dtu'])
def test_ref_nfr_training_grads_vs_oracle(setup, data_type):
    """Stage-3 training step: d loss / d (rgb_enc, diff_out, rough_out[, gamma]) through the HIP path vs the CPU oracle under
    torch autograd; the stage-2 parts (encoder, specular head) and the light are frozen (ref_nfr.py:141-146, :87)."""
    od, p, pt_, specs, m, _ = _ref_setup(data_type)
    for name in ('fine_enc', 'bottleneck', 'spec_out'):                    # what load_stage2 does to the stage-2 parts
        for prm in m.net[name].parameters():
            prm.requires_grad_(False)
    N = 200
    pts, batch, keep, ob = _ref_batch(od, N, data_type, 33, 0)
    with launches() as rec:
        pred, gt, lk, _ = m.call(batch, mode='train')
        loss, ld = m.compute_loss(pred, gt, **dict(lk))
        loss.mean().backward()
    assert rec.ran('vqn_brdf_shade_bwd')
    # round 5: rgb_enc + the two 512-wide heads on the dedicated exact-split kernels (second head input z_xyz), the frozen stage-2 parts on the
    # inference chain kernels -- nothing of the step is left on the interpreted tile programs
    assert rec.ran('vqn_refl_train_fwd_x3') and rec.ran('vqn_refl_train_bwd_x3') and not rec.ran('vqn_tile_program')
    assert rec.ran('vqn_mlp_chain_fwd')
    pt = {k: ([(od.T(W).requires_grad_(True), od.T(b).requires_grad_(True)) for W, b in v] if isinstance(v, list) else od.T(v))
          for k, v in p.items()}
    gb, gi = torch.tensor([1.3], requires_grad=True), torch.tensor([0.8], requires_grad=True)
    gamma = None if data_type == 'nerf' else od.gamma_param(gb, gi)
    want = od.ref_nfr_call(pt, specs, ob, setup['lxyz'], setup['lareas'], mode='train', data_type=data_type, gamma=gamma)
    od.ref_nfr_loss(want, ob['rgb'], data_type).mean().backward()
    np.testing.assert_allclose(_np(loss), od.ref_nfr_loss(want, ob['rgb'], data_type).detach().numpy(), rtol=1e-3, atol=2e-6)
    for name in ('rgb_enc', 'diff_out', 'rough_out'):
        for layer, (W, b) in zip(m.net[name].layers, pt[name]):
            for got, ref in ((layer.kernel.grad, W.grad), (layer.bias.grad, b.grad)):
                ref = ref.numpy()
                assert np.abs(_np(got) - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-8) + 1e-9, name
    for name in ('fine_enc', 'bottleneck', 'spec_out'):
        assert all(l.kernel.grad is None for l in m.net[name].layers), name
    assert m._light.grad is None
    if data_type != 'nerf':
        np.testing.assert_allclose(_np(m._gamma_bias.grad), gb.grad.numpy(), rtol=2e-3)
        np.testing.assert_allclose(_np(m._gamma_index.grad), gi.grad.numpy(), rtol=2e-3)


@pytest.mark.parametrize('mode,full_vis', [('test', False), ('vali', False), ('vali', True)])
@pytest.mark.parametrize('n,K', [(5000, 15), (31, 15), (4097, 15), (3000, 64), (2000, 33), (1000, 16), (1500, 17)])
def test_fused_front_is_bit_identical_to_the_separate_launches(mode, full_vis, n, K):
    """vq_nfr.Model.call in inference mode, K <= 64: encoder -> heads -> VQ step -> VQ heads as ONE launch (vqn_mlp_chain_vq_fwd, z
    and the quantised rows never leave LDS) against the four-launch path (enc + heads program, vqn_vq_quantize_rows, VQ-heads
    program): every output tensor, the indices and the lazily produced rows bit for bit; the commitment term to fp32 rounding
    (different grouping of its partial sums)."""
    from oracle import decomp as od
    from tests.decomp_util import make_config, load_oracle_params, make_batch
    from tests.gpu_util import launches
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, specs = od.make_model_params(seed=0, K=K)
    model = load_oracle_params(get_model_class('vq_nfr')(make_config(num_embed=K)), p, 'cuda')
    batch = make_batch(od.make_points(n, seed=6), 'cuda', bg_every=7)
    out = {}
    for fused in (True, False):
        model.fuse_front = fused
        with torch.no_grad(), launches() as rec:
            pred, gt, lk, tv = model.call(batch, mode=mode, full_vis=full_vis)
        assert rec.ran('vqn_mlp_chain_vq_fwd') == fused, rec.counts
        if fused:
            assert not rec.ran('vqn_vq_quantize_rows') and rec.counts.get('vqn_mlp_chain_fwd', 0) == 0, rec.counts
        out[fused] = (pred, lk, tv)
    (pa, la, ta), (pb, lb, tb) = out[True], out[False]
    assert set(pa) == set(pb)
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
    assert torch.equal(la['vqrgb'], lb['vqrgb']) and torch.equal(la['rgb'], lb['rgb'])
    np.testing.assert_allclose(float(la['vqloss']), float(lb['vqloss']), rtol=2e-6)
    assert torch.equal(la['z'], lb['z'])                                      # materialised (vali) or recomputed on access (test)
    if full_vis:
        assert torch.equal(ta['enc_z'], tb['enc_z'])


def test_fused_front_with_no_foreground_rows():
    """A view without a single foreground pixel: the one-launch front returns empty tensors (no kernel is launched), the commitment
    term is NaN like the reference's mean over nothing."""
    from oracle import decomp as od
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, specs = od.make_model_params(seed=0, K=15)
    model = load_oracle_params(get_model_class('vq_nfr')(make_config(num_embed=15)), p, 'cuda')
    batch = list(make_batch(od.make_points(64, seed=2), 'cuda'))
    batch[5] = torch.zeros_like(batch[5])                                     # alpha = 0 everywhere
    with torch.no_grad(), launches() as rec:
        pred, gt, lk, _ = model.call(tuple(batch), mode='test')
    assert rec.ran('vqn_mlp_chain_vq_fwd')
    assert pred['rgb'].shape == (64, 3) and float(pred['rgb'].abs().max()) == 0.0 and float(pred['embed'].abs().max()) == 0.0
    assert np.isnan(float(lk['vqloss']))


def test_fused_front_random_shapes():
    """A sweep over point counts (tile boundaries, single rows) and codebook sizes: one-launch front == separate launches, bit for bit."""
    from oracle import decomp as od
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    rng = np.random.default_rng(123)
    for K in (2, 9, 16, 31, 48, 64):
        p, specs = od.make_model_params(seed=int(rng.integers(100)), K=K)
        model = load_oracle_params(get_model_class('vq_nfr')(make_config(num_embed=K)), p, 'cuda')
        for n in [1, 32, 33, 63, 64, 65] + [int(v) for v in rng.integers(2, 20000, 3)]:
            batch = make_batch(od.make_points(n, seed=int(rng.integers(1000))), 'cuda', bg_every=int(rng.integers(2, 9)))
            res = {}
            for fused in (True, False):
                model.fuse_front = fused
                with torch.no_grad():
                    pred, _, lk, _ = model.call(batch, mode='test')
                res[fused] = (pred, lk)
            for k in res[True][0]:
                assert torch.equal(res[True][0][k], res[False][0][k]), (K, n, k)
            assert torch.equal(res[True][1]['vqrgb'], res[False][1]['vqrgb']), (K, n)
            a, b = float(res[True][1]['vqloss']), float(res[False][1]['vqloss'])
            assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 2e-6 * max(abs(b), 1e-30), (K, n, a, b)


def test_fused_srgb_transfer_matches_the_torch_statement():
    """vqn_linear2srgb (one pass) against clamp -> where(t <= 0.0031308, 12.92 t, 1.055 t^(1/2.4) - 0.055) in torch: values outside
    [0, 1], the knee, exact 0 and 1."""
    from vqnerf_release_amd import _C
    from vqnerf_release_amd.decomp.nerfactor.util import img as imgutil
    g = torch.Generator(device='cuda'); g.manual_seed(5)
    x = torch.cat([torch.rand(100003, device='cuda', generator=g) * 1.4 - 0.2,
                   torch.tensor([0.0, 1.0, 0.0031308, 0.0031307, 0.0031309, -1.0, 2.0, 1e-8], device='cuda')]).reshape(-1, 3)
    got = _C.linear2srgb(x)
    with torch.enable_grad():
        want = imgutil.linear2srgb(x.clone().requires_grad_(True)).detach()     # the torch statement (autograd path)
    np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=2e-7)
    assert torch.equal(imgutil.linear2srgb(x), got)                              # the inference path dispatches to the kernel
    # NaN in -> NaN out, like the clip of the statement (a rendered NaN must not turn into a black pixel); +-inf clip to 1 / 0
    bad = torch.tensor([float('nan'), 0.5, float('inf'), -float('inf'), float('nan'), 0.25], device='cuda')
    got_b = _C.linear2srgb(bad)
    with torch.enable_grad():
        want_b = imgutil.linear2srgb(bad.clone().requires_grad_(True)).detach()
    assert torch.isnan(got_b[0]) and torch.isnan(got_b[4]) and torch.equal(torch.isnan(got_b), torch.isnan(want_b))
    ok = ~torch.isnan(want_b)
    np.testing.assert_allclose(got_b[ok].cpu().numpy(), want_b[ok].cpu().numpy(), rtol=0, atol=2e-7)


@pytest.mark.parametrize('data_type', ['nerf', 'hw'])
@pytest.mark.parametrize('N', [2, 258, 4096])
def test_fused_train_loss_matches_the_torch_statement(data_type, N):
    """`FusedTrainLoss` (vqn_decomp_loss_fwd / _bwd: the per-point terms of compute_loss's train branch, vq_nfr.py:906-981, two
    launches) against the torch statement of the same branch (`fuse_train_loss = False`), values and gradients -- incl. rows with
    vq_rgb = 0 (TensorFlow's divide_no_nan / SqrtGrad give a zero gradient there; the torch statement gives NaN, so those rows are
    compared with 0), chromaticity differences on both sides of `chr_thres`, rough on both sides of 0.5, ties in max(spec)."""
    from tests.decomp_util import make_config
    from tests.gpu_util import launches
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    model = get_model_class('vq_nfr')(make_config(data_type=data_type, num_embed=8))
    model.build_nets(device='cuda', seed=0).to('cuda')
    rng = np.random.default_rng(N)
    model.set_codebook(rng.uniform(0, 1, (8, 256)).astype(np.float32))
    T = lambda a: torch.tensor(a.astype(np.float32), device='cuda')
    gt = rng.uniform(0.02, 1, (N, 3))
    gt[1::2] = np.where(rng.uniform(size=(N // 2, 1)) < 0.5, gt[::2] + rng.normal(0, 0.01, (N // 2, 3)), gt[1::2]).clip(0.02, 1)   # pairs under the threshold
    vals = dict(rgb=T(rng.uniform(0, 1, (N, 3))), vqrgb=T(rng.uniform(0, 1, (N, 3))), z=T(rng.normal(size=(N, 256)) / 16),
                spec=T(rng.uniform(0, 0.3, (N, 3))), rough=T(rng.uniform(0, 1, (N, 1))))
    zero_rows = np.zeros(N, bool)
    if N > 2:
        zero_rows[[3, 10]] = True
        vals['vqrgb'][zero_rows] = 0.0
        vals['spec'][5] = 0.2                                              # a three-way tie in max(spec)
    rgb_gt = T(gt)
    out = {}
    for fused in (False, True):
        model.fuse_train_loss = fused
        leaves = {k: v.clone().requires_grad_(True) for k, v in vals.items()}
        kw = dict(mode='train', gtc=rgb_gt, rgb=leaves['rgb'], vqrgb=leaves['vqrgb'], vqloss=torch.tensor(0.37, device='cuda'), z=leaves['z'],
                  spec=leaves['spec'], rough=leaves['rough'], embed=model._codebook)
        with launches() as rec:
            loss, ld = model.compute_loss({}, {}, **kw)
            wts = T(rng.uniform(0.5, 1.5, (N,))) if fused is None else torch.linspace(0.5, 1.5, N, device='cuda')
            (loss * wts).sum().backward()
        assert rec.ran('vqn_decomp_loss_fwd') == fused and rec.ran('vqn_decomp_loss_bwd') == fused
        out[fused] = (loss.detach(), {k: v.detach() for k, v in ld.items()}, {k: v.grad for k, v in leaves.items()})
    (l0, d0, g0), (l1, d1, g1) = out[False], out[True]
    assert set(d0) == set(d1) == {'rgb', 'vqrgb', 'vqloss', 'chromaticity', 'chr_smooth', 'sim_smooth', 'lambert', 'loss'}
    np.testing.assert_allclose(l1.cpu().numpy(), l0.cpu().numpy(), rtol=2e-5, atol=1e-7)
    for k in d0:
        np.testing.assert_allclose(d1[k].cpu().numpy(), d0[k].cpu().numpy(), rtol=2e-5, atol=1e-7, err_msg=k)
    ok = ~torch.tensor(zero_rows).cuda()
    for k in ('rgb', 'vqrgb', 'z', 'spec'):
        a, b = g1[k], g0[k]
        assert torch.isfinite(a).all(), k
        scale = float(b[ok].abs().max())
        np.testing.assert_allclose(a[ok].cpu().numpy(), b[ok].cpu().numpy(), rtol=1e-4, atol=1e-6 * max(scale, 1e-3), err_msg=k)
    assert g1['rough'] is None or float(g1['rough'].abs().max()) == 0.0      # rough is detached in the lambert term (vq_nfr.py:973)
    if zero_rows.any():
        # chromaticity has no gradient where |vq_rgb| = 0 (the mse term's is still there)
        lin = (torch.where(rgb_gt <= 0.04045, rgb_gt / 12.92, ((rgb_gt + 0.055) / 1.055) ** 2.4) if data_type == 'nerf' else rgb_gt)
        want = (2.0 / 3.0) * (vals['vqrgb'] - lin) * torch.linspace(0.5, 1.5, N, device='cuda')[:, None]
        np.testing.assert_allclose(g1['vqrgb'][~ok].cpu().numpy(), want[~ok].cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_codebook_prep_and_code_separation_kernels_match_the_torch_statements():
    """vqn_codebook_prep / vqn_sim_smooth_* (one launch each way) against autograd over the torch statements of get_codebook()
    (clip with identity gradient + l2-normalise per code) and of the code-separation term (vq_nfr.py:955-968): values and the
    gradient wrt the raw codebook variable, entries outside [0, 1] and a code below the eps clamp included."""
    from vqnerf_release_amd.decomp.nerfactor.models import vq_nfr
    from vqnerf_release_amd.decomp.nerfactor.util import math as mathutil
    for D, K in ((256, 15), (256, 64), (32, 2)):
        g = torch.Generator(device='cuda').manual_seed(K)
        raw = torch.rand(D, K, device='cuda', generator=g) * 1.4 - 0.2            # some entries outside [0, 1]
        raw[:, 0] *= 1e-5                                                          # sum c^2 < eps: the clamp of the normalisation is active
        gy = torch.randn(D, K, device='cuda', generator=g)
        w = 0.3

        def torch_statement(x):
            cb = mathutil.safe_l2_normalize(mathutil.clip_preserve_gradient(x, 0.0, 1.0), axis=0)
            c = cb.t()
            eye = torch.eye(K, device='cuda')
            dist = torch.sqrt(((c[:, None, :] - c[None, :, :]) ** 2).sum(-1) + eye) * (1 - eye)
            masked = dist * (1 - eye) + eye * dist.max()
            return cb, w * (-torch.log(masked.min()))

        xr = raw.clone().requires_grad_(True)
        cb_ref, sim_ref = torch_statement(xr)
        ((cb_ref * gy).sum() + 2.0 * sim_ref).backward()
        xh = raw.clone().requires_grad_(True)
        cb = vq_nfr._CodebookPrep.apply(xh)
        sim = vq_nfr._SimSmooth.apply(cb, w)
        ((cb * gy).sum() + 2.0 * sim).backward()
        # (the term is -w log(d) with d near 1 for two codes: its error is absolute, ~ w ulp(d))
        assert float((cb - cb_ref).detach().abs().max()) <= 2e-7 and abs(float(sim.detach()) - float(sim_ref.detach())) <= 2e-6 * max(abs(float(sim_ref.detach())), w)
        assert float((xh.grad - xr.grad).abs().max()) <= 2e-5 * float(xr.grad.abs().max())


def test_ref_nfr_with_a_trainable_encoder_keeps_the_interpreted_programs(setup):
    """Stage 3 with NOTHING frozen (not the reference's schedule -- load_stage2 freezes the stage-2 parts -- but a legal model state): z_xyz
    wants an adjoint, which the dedicated second-input kernel does not produce, so the step takes the interpreted programs for the
    512-wide heads and every network, the encoder included, receives a gradient that agrees with the oracle under torch autograd."""
    od, p, pt_, specs, m, _ = _ref_setup('nerf')
    N = 150
    pts, batch, keep, ob = _ref_batch(od, N, 'nerf', 35, 0)
    with launches() as rec:
        pred, gt, lk, _ = m.call(batch, mode='train')
        loss, ld = m.compute_loss(pred, gt, **dict(lk))
        loss.mean().backward()
    assert rec.ran('vqn_tile_program')
    pt = {k: ([(od.T(W).requires_grad_(True), od.T(b).requires_grad_(True)) for W, b in v] if isinstance(v, list) else od.T(v))
          for k, v in p.items()}
    want = od.ref_nfr_call(pt, specs, ob, setup['lxyz'], setup['lareas'], mode='train', data_type='nerf', gamma=None)
    od.ref_nfr_loss(want, ob['rgb'], 'nerf').mean().backward()
    for name in ('fine_enc', 'bottleneck', 'spec_out', 'rgb_enc', 'diff_out', 'rough_out'):
        for layer, (W, b) in zip(m.net[name].layers, pt[name]):
            for got, ref in ((layer.kernel.grad, W.grad), (layer.bias.grad, b.grad)):
                ref = ref.numpy()
                assert got is not None and np.abs(_np(got) - ref).max() <= 2e-3 * max(np.abs(ref).max(), 1e-8) + 1e-9, name
