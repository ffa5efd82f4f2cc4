"""CPU: host logic of the geo half -- the HOCON-subset conf parser against the reference's conf grammar, the
runner's schedules and checkpoint keys."""
import glob
import os

import numpy as np
import pytest
import torch

from vqnerf_release_amd.geo import conf as hocon

HERE = os.path.dirname(os.path.abspath(__file__))


def test_conf_grammar():
    c = hocon.parse_file(os.path.join(HERE, 'golden', 'neus_like.conf'), case='lego')
    assert c['general.base_exp_dir'] == './exp/lego/nerf' and c['general.recording'] == ['./', './models']
    assert c.get_int('train.batch_size') == 64 and c.get_float('train.learning_rate') == 5e-4
    assert c.get_bool('train.use_white_bkgd') is True and c.get_float('train.anneal_end', default=0.0) == 0.0
    assert c.get_float('train.not_there', default=1.5) == 1.5
    with pytest.raises(KeyError):
        c.get_int('train.not_there')
    assert c['model.sdf_network'] == dict(d_out=65, d_in=3, d_hidden=64, n_layers=4, skip_in=[2], multires=6, bias=0.5,
                                          scale=1.0, geometric_init=True, weight_norm=True)
    assert c['model.nerf']['skips'] == [4] and c['model.nerf']['use_viewdirs'] is True
    assert c['model.neus_renderer']['up_sample_steps'] == 4          # trailing comment stripped
    assert c['dataset.near'] == 2.0
    c['dataset.data_dir'] = c['dataset.data_dir'].replace('nfr_blender', 'x')      # nerf_runner.py:33 assigns by dotted key
    assert c['dataset']['data_dir'] == './data/x/lego/'
    t = hocon.parse_string('a { b = 1, c : "q r" }\na.d = [1, 2,\n 3]\n// note\ne = off')
    assert t['a'] == {'b': 1, 'c': 'q r', 'd': [1, 2, 3]} and t['e'] is False


@pytest.mark.skipif(not os.path.isdir('/root/reference/geo/NeuS-ours2/confs'), reason='reference tree not present')
def test_every_reference_conf_parses():
    files = sorted(glob.glob('/root/reference/geo/NeuS-ours2/confs/**/*.conf', recursive=True))
    assert len(files) == 10
    for f in files:
        c = hocon.parse_file(f, case='scene')
        assert c['model.sdf_network']['d_hidden'] == 256 and c['model.neus_renderer']['n_outside'] == 0
        assert c.get_int('train.batch_size') in (512, 2560, 5120)
        from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork
        SingleVarianceNetwork(**c['model.variance_network'])


def test_runner_schedules_and_checkpoint_keys(tmp_path):
    from vqnerf_release_amd.geo.nerf_runner import Runner, SyntheticDataset
    text = open(os.path.join(HERE, 'golden', 'neus_like.conf')).read().replace('./exp/', str(tmp_path) + '/exp/')
    r = Runner(conf_text=text, case='lego', device='cpu', dataset=SyntheticDataset(device='cpu', n_images=4, H=32, W=32))
    r.update_learning_rate()
    assert r.optimizer.param_groups[0]['lr'] == 0.0                         # warm-up starts at 0 (nerf_runner.py:187-188)
    r.iter_step = 2500; r.update_learning_rate()
    assert abs(r.optimizer.param_groups[0]['lr'] - 2.5e-4) < 1e-12
    r.iter_step = 300000; r.update_learning_rate()
    assert abs(r.optimizer.param_groups[0]['lr'] - 5e-4 * 0.05) < 1e-12      # cosine floor = alpha
    assert r.get_cos_anneal_ratio() == 1.0
    r.iter_step = 7
    r.save_checkpoint()
    ck = torch.load(os.path.join(r.base_exp_dir, 'checkpoints', 'ckpt_000007.pth'), weights_only=False)
    assert set(ck) == {'nerf', 'sdf_network_fine', 'variance_network_fine', 'color_network_fine', 'optimizer', 'iter_step'}
    assert {'lin0.weight_g', 'lin0.weight_v', 'lin0.bias'} <= set(ck['sdf_network_fine']) and 'variance' in ck['variance_network_fine']
    r2 = Runner(conf_text=text, case='lego', device='cpu', is_continue=True,
                dataset=SyntheticDataset(device='cpu', n_images=4, H=32, W=32))
    assert r2.iter_step == 7
    d = r.dataset.gen_random_rays_at(1, 50)
    assert d.shape == (50, 10) and torch.allclose(d[:, 3:6].norm(dim=-1), torch.ones(50), atol=1e-6)


def test_bench_reads_the_profiler_counter_files(tmp_path):
    """bench.py's live `roofline.traffic`: the parser of rocprofv3's per-dispatch counter CSV (kernel names carry template spaces)."""
    import bench
    d = tmp_path / 'x' / 'y'
    d.mkdir(parents=True)
    (d / 'c_counter_collection.csv').write_text(
        '"Correlation_Id","Kernel_Name","Counter_Name","Counter_Value"\n'
        '1,"void (anonymous namespace)::neus_points2_kernel<true>(SdfDesc, ColDesc)","FETCH_SIZE",1000.5\n'
        '2,"void (anonymous namespace)::neus_points2_kernel<false>(SdfDesc, ColDesc)","FETCH_SIZE",7\n'
        '3,"void (anonymous namespace)::neus_points2_kernel< true >(SdfDesc, ColDesc)","FETCH_SIZE",999.5\n'
        '4,"void (anonymous namespace)::neus_points2_kernel<true>(SdfDesc, ColDesc)","WRITE_SIZE",5\n')
    assert bench.pmc_values(str(tmp_path), 'FETCH_SIZE', 'neus_points2_kernel<true>') == [1000.5, 999.5]
    # the name bench looks for is the fine render kernel's, not the training instantiation's
    (tmp_path / 'p').mkdir()
    (tmp_path / 'p' / 'd_counter_collection.csv').write_text(
        'Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n'
        '1,"void (anonymous namespace)::neus_points2_kernel<true, false>(SdfDesc, ColDesc, TrainOut)","FETCH_SIZE",11\n'
        '2,"void (anonymous namespace)::neus_points2_kernel<true, true>(SdfDesc, ColDesc, TrainOut)","FETCH_SIZE",13\n')
    assert bench.pmc_values(str(tmp_path), 'FETCH_SIZE', bench.FINE_KERNEL) == [11.0]
    assert bench.pmc_values(str(tmp_path), 'WRITE_SIZE', 'neus_points2_kernel<true>') == [5.0]
    assert bench.pmc_values(str(tmp_path), 'FETCH_SIZE', 'no_such_kernel') == []


def test_fused_forward_pack_indices_reproduce_the_render_packs():
    """The training forward on the render kernel (vqn_neus_train_fwd) takes its two weight packs as ONE gather each from the
    flat source vector of NeusTrainEngine.pack(); those gather indices must give exactly SdfPackPlan / ColPackPlan.pack() of the
    same effective weights (skip layer's 1/sqrt2 included) and the same descriptors."""
    import math
    from vqnerf_release_amd.geo import packing
    from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork
    from vqnerf_release_amd.geo.train_programs import NeusTrainEngine
    sdf = SDFNetwork(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1.5, geometric_init=True,
                     weight_norm=True)
    col = RenderingNetwork(d_feature=256, mode='idr', d_in=9, d_out=3, d_hidden=256, n_layers=4, weight_norm=True, multires_view=4,
                           squeeze_out=True)
    e = NeusTrainEngine(sdf, col)
    assert e.fused_forward()
    gi_s, d_s, gi_c, d_c = e._fused_static(torch.device('cpu'))
    g = torch.Generator().manual_seed(0)
    W = [torch.randn(e.out[l], e.inn[l], generator=g) for l in range(e.nL + 1)]
    b = [torch.randn(e.out[l], generator=g) for l in range(e.nL + 1)]
    Wc = [torch.randn(e.cout[l], e.cin[l], generator=g) for l in range(e.nC + 1)]
    bc = [torch.randn(e.cout[l], generator=g) for l in range(e.nC + 1)]
    src = {}
    for l in range(e.nL + 1):
        src['W%d' % l], src['b%d' % l] = (W[l] / math.sqrt(2.0) if l == e.skip else W[l]), b[l]
    for l in range(e.nC + 1):
        src['Wc%d' % l], src['bc%d' % l] = Wc[l], bc[l]
    flat = e._layout().flatten(src)
    plan = sdf.plan(max_tiles=col.max_tiles())
    wb, d = plan.pack(W, b)
    assert torch.equal(wb, flat[gi_s]) and (d == d_s).all()
    cp = packing.ColPackPlan(col.d_feature, col.mode, col.dims[1], col.num_layers - 2, col.dims[-1], col.multires_view, col.squeeze_out,
                             plan.tiles[-1])
    wbc, dc = cp.pack(Wc, bc)
    assert torch.equal(wbc, flat[gi_c]) and (dc == d_c).all()
    # narrow networks (fewer than five feature tiles) keep the interpreted forward
    small = NeusTrainEngine(SDFNetwork(d_in=3, d_out=65, d_hidden=64, n_layers=4, skip_in=(2,), multires=6, weight_norm=True),
                            RenderingNetwork(d_feature=64, mode='idr', d_in=9, d_out=3, d_hidden=64, n_layers=2, weight_norm=True,
                                             multires_view=4, squeeze_out=True))
    assert not small.fused_forward()
