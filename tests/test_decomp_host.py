"""CPU: host logic of the reflectance models -- config surface, registry, Keras-layout Dense stacks, the torch
statements used by the autograd path (BRDF, rendering sum, loss) against oracle/decomp.py, and the rule that the
no-graph path never falls back to the CPU."""
import os

import numpy as np
import pytest
import torch

from oracle import decomp as od
from tests.decomp_util import make_config, load_oracle_params, make_batch
from vqnerf_release_amd import _C
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util import config as configutil, microfacet


@pytest.fixture(scope='module')
def setup():
    p, specs = od.make_model_params(seed=0, K=15)
    model = load_oracle_params(get_model_class('vq_nfr')(make_config()), p, 'cpu')
    pt = {k: ([(od.T(W), od.T(b)) for W, b in v] if isinstance(v, list) else od.T(v)) for k, v in p.items()}
    return dict(p=p, pt=pt, specs=specs, model=model)


def test_registry_and_config_surface():
    cfg = make_config()
    configutil.apply_override(cfg, 'num_embed=8,thres_str=0.1;0.2;0.3,lr=1e-3')
    assert cfg.getint('DEFAULT', 'num_embed') == 8 and cfg.get('DEFAULT', 'thres_str').split(';') == ['0.1', '0.2', '0.3']
    m = get_model_class('vq_nfr')(cfg)
    assert m.num_embed == 8 and m.vq_layer.num_embeddings == 8 and m.light_res == (16, 32)
    assert set(m.net) == {'fine_enc', 'bottleneck', 'diff_main', 'spec_main', 'rough_main', 'diff_vq', 'spec_vq', 'rough_vq'}
    m.build_nets(seed=1)
    assert m.trainable_registered and hasattr(m, 'net_fine_enc_layer0')
    n_params = sum(p.numel() for p in m.trainable_variables)
    assert n_params == 179968 + 296832 + 297600 + sum(sum(n.widths) for n in m.net.values())     # MACs + biases
    with pytest.raises(ValueError):
        m._validate_mode('bogus')
    assert configutil.get_config_ini('/out/train/scene/lr5e-4/checkpoints/ckpt-5') == '/out/train/scene/lr5e-4.ini'
    assert get_model_class('nfr_unit')(make_config()).net.keys() >= {'diff_out', 'spec_out', 'rough_out', 'fine_enc', 'bottleneck'}


def test_dense_stacks_match_oracle(setup):
    model, pt, specs = setup['model'], setup['pt'], setup['specs']
    xyz = od.T(od.make_points(50, seed=2)['xyz']).requires_grad_(True)      # graph needed -> torch statements
    z = model._pred_enc_at(xyz)
    torch.testing.assert_close(z, od.pred_enc(pt, specs, xyz.detach()), rtol=0, atol=1e-6)
    for vq in (False, True):
        got = model._all_heads(z, 'vq' if vq else 'main')
        for g, w in zip(got, od.heads(pt, specs, z.detach(), vq)):
            torch.testing.assert_close(g, w, rtol=0, atol=1e-6)


def test_no_graph_path_refuses_cpu_tensors(setup):
    model = setup['model']
    with torch.no_grad():
        with pytest.raises(_C.VqnError, match='no CPU fallback'):
            model._pred_enc_at(torch.zeros(4, 3))
    from vqnerf_release_amd.geo.models.fields import SDFNetwork
    sdf = SDFNetwork(d_in=3, d_out=65, d_hidden=64, n_layers=2, skip_in=(), multires=6)
    with torch.no_grad():
        with pytest.raises(_C.VqnError, match='no CPU fallback'):
            sdf.sdf(torch.zeros(4, 3))


def test_brdf_and_render_statements_match_oracle(setup):
    model = setup['model']
    N = 40
    pts = od.make_points(N, seed=4)
    rng = np.random.default_rng(5)
    a, s, r = od.T(rng.uniform(0, 1, (N, 3))), od.T(rng.uniform(0, 1, (N, 3))), od.T(rng.uniform(0, 1, (N, 1)))
    xyz, normal, rayo, lvis = od.T(pts['xyz']), od.T(pts['normal']), od.T(pts['rayo']), od.T(pts['lvis'])
    l = model._calc_ldir(xyz)
    v = model._calc_vdir(rayo, xyz)
    n = model._normal_correct(normal, v)
    lxyz, lareas = od.gen_light_xyz(16, 32)
    torch.testing.assert_close(l, od.calc_ldir(od.T(lxyz), xyz), rtol=0, atol=0)
    got = microfacet.get_brdf(l, v, n, albedo=a, rough=r, f0=s)
    want = od.get_brdf(l, v, n, a, r, s)
    for g, w in zip(got, want):
        torch.testing.assert_close(g, w, rtol=1e-6, atol=1e-7)
    rgb, _, _ = model._render(got[0], l, n, lvis)
    torch.testing.assert_close(rgb, od.render_integrate(want[0], l, n, od.T(lareas), model.light.detach(), lvis), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('mode', ['train', 'vali'])
def test_compute_loss_matches_oracle(setup, mode):
    model, pt = setup['model'], setup['pt']
    N = 64
    rng = np.random.default_rng(6)
    U = lambda *s: od.T(rng.uniform(0, 1, s))
    z_vq = od.safe_l2_normalize(U(N, 256), 1)
    out = dict(rgb=U(N, 3), vq_rgb=U(N, 3), z_vq=z_vq, spec=U(N, 3), rough=U(N, 1), vq=dict(loss=torch.tensor(0.0123)))
    rgb_gt = U(N, 3)
    want, wd = od.compute_loss(out, rgb_gt, pt['codebook_raw'], mode=mode)
    lk = {'vqloss': out['vq']['loss'], 'vqrgb': out['vq_rgb'], 'mode': mode, 'gtc': rgb_gt, 'rgb': out['rgb'],
          'spec': out['spec'], 'rough': out['rough'], 'z': z_vq, 'embed': model._codebook}
    got, gd = model.compute_loss({}, {}, **lk)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7)
    assert set(gd) == set(wd)


def test_hdr_probe_reader_and_novel_lights(tmp_path):
    """Radiance RGBE probes (the reference reads them through xiuminglib / OpenCV): flat and run-length-encoded scanlines,
    value = mantissa * 2^(e - 136); `test_envmap_dir` -> model.novel_probes, plus the four OLAT maps of nfr_unit.py:66-79."""
    from vqnerf_release_amd.decomp.nerfactor.util import io as ioutil
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    rng = np.random.default_rng(0)
    arr = (rng.uniform(0, 1, (16, 32, 3)) * 10 ** rng.uniform(-3, 2, (16, 32, 1))).astype(np.float32)
    arr[3, 5] = 0.0
    ioutil.write_hdr(str(tmp_path / 'b_city.hdr'), arr)
    back = ioutil.read_hdr(str(tmp_path / 'b_city.hdr'))
    assert back.shape == (16, 32, 3) and (back[3, 5] == 0).all()
    np.testing.assert_allclose(back, arr, rtol=0, atol=float(arr.max(-1).max()) / 128)      # 8-bit mantissa of the brightest channel
    np.testing.assert_allclose(back.max(-1), arr.max(-1), rtol=1 / 128)
    # the same pixels, run-length encoded by hand: per scanline 2 2 W_hi W_lo, then per channel (128 + run, value) / (n, literals)
    flat = open(tmp_path / 'b_city.hdr', 'rb').read()
    head_end = flat.index(b'+X 32\n') + 6
    rgbe = np.frombuffer(flat[head_end:], np.uint8).reshape(16, 32, 4)
    body = bytearray()
    for y in range(16):
        body += bytes([2, 2, 0, 32])
        for c in range(4):
            row = rgbe[y, :, c]
            if c == 3 and (row == row[0]).all():
                body += bytes([128 + 32, int(row[0])])
            else:
                body += bytes([20]) + row[:20].tobytes() + bytes([12]) + row[20:].tobytes()
    open(tmp_path / 'a_rle.hdr', 'wb').write(flat[:head_end] + bytes(body))
    np.testing.assert_array_equal(ioutil.read_hdr(str(tmp_path / 'a_rle.hdr')), back)
    np.save(tmp_path / 'c_big.npy', rng.uniform(0, 1, (32, 64, 3)).astype(np.float32))    # resized to the 16 x 32 light grid
    with pytest.raises(ValueError):
        open(tmp_path / 'bad.hdr', 'wb').write(b'P6\n1 1\n255\n')
        ioutil.read_hdr(str(tmp_path / 'bad.hdr'))
    os.remove(tmp_path / 'bad.hdr')
    m = get_model_class('vq_nfr')(make_config(test_envmap_dir=str(tmp_path), olat_inten=200, ambient_inten=0.5))
    assert list(m.novel_probes) == ['a_rle', 'b_city', 'c_big'] and all(v.shape == (16, 32, 3) for v in m.novel_probes.values())
    np.testing.assert_array_equal(m.novel_probes['b_city'].numpy(), back)
    assert list(m.novel_olat) == ['0004-0000', '0004-0008', '0004-0016', '0004-0024']
    o = m.novel_olat['0004-0008']
    assert float(o[4, 8, 0]) == 200.5 and float(o[0, 0, 0]) == 0.5 and float(o.sum()) == pytest.approx(3 * (200 + 0.5 * 512))
    m2 = m.double()                                                 # _apply reaches the maps
    assert m2.novel_probes['c_big'].dtype == torch.float64


def test_check_numerics_guards_are_optional_debug_checks(setup):
    """The reference's tf.debugging.check_numerics sites (vq_nfr.py:783 "Z", :802 "Albedo", :815, :827, :985 "Loss"): off by
    default (each is a host sync), on with Model(config, debug=True) or VQN_CHECK_NUMERICS=1, raising like TF does."""
    from vqnerf_release_amd.decomp.nerfactor.util.math import InvalidArgumentError, check_numerics
    assert check_numerics(torch.ones(3), 'x') is not None
    with pytest.raises(InvalidArgumentError, match='Loss : Tensor had NaN'):
        check_numerics(torch.tensor([1.0, float('nan')]), 'Loss')
    with pytest.raises(InvalidArgumentError, match='Inf'):
        check_numerics(torch.tensor([float('inf')]), 'Z')
    p = setup['p']
    quiet = load_oracle_params(get_model_class('vq_nfr')(make_config()), p, 'cpu')
    loud = load_oracle_params(get_model_class('vq_nfr')(make_config(), debug=True), p, 'cpu')
    assert not quiet.check_numerics and loud.check_numerics
    xyz = torch.tensor(od.make_points(8, seed=1)['xyz'])
    for m in (quiet, loud):
        with torch.no_grad():
            m.net['fine_enc'].layers[0].kernel[0, 0] = float('nan')
    assert torch.isnan(quiet._pred_enc_at(xyz)).any()                   # silently propagates, as any torch op would
    with pytest.raises(InvalidArgumentError, match='^Z : '):
        loud._pred_enc_at(xyz)
    with pytest.raises(InvalidArgumentError, match='^Albedo : '):
        loud._pred_diff_at(torch.full((4, 256), float('nan')))


def test_stage3_stack_descriptor_is_accepted_by_the_library():
    """CPU: the descriptor + gather indices `ReflStackEngine(zx=True)` lays out for the stage-3 stack (rgb_enc + the two 512-wide heads,
    ref_nfr.py:137-152) -- shapes of the layout, the second-input fields, and the library's own validation of the descriptor
    (vqn_refl_train_bwd_x3_scratch_bytes returns -1 for a descriptor load_desc_r rejects)."""
    import ctypes
    from vqnerf_release_amd import _C
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    from vqnerf_release_amd.decomp.refl_train import ReflStackEngine
    m = get_model_class('ref_nfr')(make_config(model='ref_nfr'))
    m.build_nets(device='cpu', seed=1)
    enc, heads = [m.net['rgb_enc']], [m.net['diff_out'], m.net['rough_out']]
    assert ReflStackEngine.supports(enc, heads, m.z_dim, 3, zx=True)
    eng = ReflStackEngine(enc, 0, heads, m.z_dim, 'cpu', zx=True)
    L, gidx, n_steps, fidx, d = eng._static()
    shapes = dict(zip(L.names, [tuple(p.shape) for p in eng.params()]))
    assert shapes['W0'] == (3, 256) and shapes['H0_W0'] == (512, 256) and shapes['H0_W2'] == (640, 3) and shapes['H1_W2'] == (640, 1)
    assert d[0] == 3 and d[3] == 3 and d[6] == 2 and d[7] == 8 and d[8] == 256 and d[9] == 8          # n_enc, emb_feats, n_heads, z_tiles, z_feats, zx_tiles
    assert d[10] > 0 and d[11] > d[10]                                                             # offW2zx of the two heads
    assert gidx.numel() == n_steps * 512 and int(gidx.max()) <= L.zero
    lib = _C.lib()
    lib.vqn_refl_train_bwd_x3_scratch_bytes.restype = ctypes.c_int64
    dd, dp = _C._i32(d)
    assert lib.vqn_refl_train_bwd_x3_scratch_bytes(dp) > 0
    bad = d.copy(); bad[9] = 4                                                                      # zx_tiles != z_tiles
    dd2, dp2 = _C._i32(bad)
    assert lib.vqn_refl_train_bwd_x3_scratch_bytes(dp2) == -1
