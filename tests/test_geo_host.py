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
    assert bench.pmc_values(str(tmp_path), 'WRITE_SIZE', 'neus_points2_kernel<true>') == [5.0]
    assert bench.pmc_values(str(tmp_path), 'FETCH_SIZE', 'no_such_kernel') == []
