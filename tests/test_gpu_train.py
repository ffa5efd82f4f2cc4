"""GPU: the two trainers step end to end on the device -- up-sampling kernels, forward / backward tile programs
(`vqn_tile_program`), weight-gradient contraction (`vqn_wgrad_partials`), compositing and shading forward / backward kernels,
one flat gradient bucket, Adam -- and the fused inference path sees the updated weights (pack caches are invalidated by the
optimiser's in-place updates).  The HIP training engines are compared with torch autograd over the torch statements of the
same modules (`train_backend = 'torch'`); every such comparison asserts which kernels each side launched."""
import os

import numpy as np
import pytest
import torch

from tests.decomp_util import make_config, load_oracle_params, make_batch
from tests.gpu_util import launches

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_geo_runner_trains(tmp_path):
    from vqnerf_release_amd.geo.nerf_runner import Runner, SyntheticDataset
    text = open(os.path.join(HERE, 'golden', 'neus_like.conf')).read().replace('./exp/', str(tmp_path) + '/exp/')
    text = text.replace('warm_up_end = 5000', 'warm_up_end = 0').replace('batch_size = 64', 'batch_size = 256')
    torch.manual_seed(0)
    r = Runner(conf_text=text, case='lego', dataset=SyntheticDataset(n_images=4, H=64, W=64))
    probe = torch.tensor(np.random.default_rng(0).uniform(-0.8, 0.8, (64, 3)).astype(np.float32)).cuda()
    with torch.no_grad():
        sdf0 = r.sdf_network.sdf(probe).clone()                    # fused HIP path
    r.update_learning_rate()
    losses = []
    for it in range(12):
        data = r.dataset.gen_random_rays_at(it % 4, r.batch_size)
        st = r.train_step(data)
        losses.append(float(st['loss']))
    assert all(np.isfinite(losses)) and r.iter_step == 12
    assert np.mean(losses[-4:]) < np.mean(losses[:4])             # it learns
    with torch.no_grad():
        sdf1 = r.sdf_network.sdf(probe)
        ref = r.sdf_network.forward(probe)[:, :1]                  # torch statement with the updated weights
    assert not torch.equal(sdf0, sdf1)
    np.testing.assert_allclose(sdf1.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=2e-5)
    # gradients landed in the flat bucket (no copies)
    assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(r.bucket.params, r.bucket.views))


def test_geo_runner_graph_replays_the_eager_step(tmp_path):
    """Runner(graph=True): the captured optimisation step (up-sampling passes, forward / backward tile programs, compositing,
    weight-gradient contractions, weight-norm chain rule, Adam) replayed on new batches is the eager step bit for bit -- both
    sides run the same capturable Adam; `perturb = 0` so that no random draw enters (with draws the replays take theirs from the
    device generator: checked for finiteness and learning below)."""
    from vqnerf_release_amd.geo.nerf_runner import Runner, SyntheticDataset
    from tests.gpu_util import launches
    text = open(os.path.join(HERE, 'golden', 'neus_like.conf')).read().replace('./exp/', str(tmp_path) + '/exp/')
    text = text.replace('warm_up_end = 5000', 'warm_up_end = 0').replace('batch_size = 64', 'batch_size = 256')
    assert 'perturb = 1.0' in text and 'anneal_end' in text
    det = text.replace('perturb = 1.0', 'perturb = 0.0')
    import re
    det = re.sub(r'anneal_end = [0-9.]+', 'anneal_end = 0', det)
    runs = {}
    for graph in (False, True):
        torch.manual_seed(0)
        r = Runner(conf_text=det, case='lego', dataset=SyntheticDataset(n_images=4, H=64, W=64), graph=True)
        r.graph = graph                                           # same (capturable, fused) Adam on both sides; eager when False
        r.update_learning_rate()
        torch.manual_seed(1)
        batches = [r.dataset.gen_random_rays_at(it % 4, r.batch_size) for it in range(8)]
        losses = []
        with launches() as rec:
            for b in batches:
                losses.append(r.train_step(b)['loss'].clone())
        assert (r._cap is not None) == graph and r.iter_step == 8
        assert (rec.ran('vqn_tile_program:prog_sbwd') or rec.ran('vqn_neus_train_bwd')) and rec.ran('vqn_wgrad_partials')
        runs[graph] = ([float(x) for x in losses], [p.detach().clone() for p in r.bucket.params],
                       float(r.optimizer.param_groups[0]['lr']))
    (l0, w0, lr0), (l1, w1, lr1) = runs[False], runs[True]
    assert l0 == l1, (l0, l1)
    assert all(torch.equal(a, b) for a, b in zip(w0, w1)) and lr0 == lr1
    assert l1[-1] != l1[-2]                                       # the replays consumed different batches
    # inference after replays sees the moved weights (pack caches key on the weights epoch)
    probe = torch.tensor(np.random.default_rng(0).uniform(-0.8, 0.8, (64, 3)).astype(np.float32)).cuda()
    with torch.no_grad():
        np.testing.assert_allclose(r.sdf_network.sdf(probe).cpu().numpy(), r.sdf_network.forward(probe)[:, :1].cpu().numpy(), rtol=0, atol=2e-5)
    with pytest.raises(ValueError):
        r.train_step(batches[0][:100])
    # with the random perturbation of the shipped conf: draws come from the device generator inside the graph
    torch.manual_seed(0)
    r2 = Runner(conf_text=text, case='lego', dataset=SyntheticDataset(n_images=4, H=64, W=64), graph=True)
    r2.update_learning_rate()
    ls = [float(r2.train_step(r2.dataset.gen_random_rays_at(it % 4, r2.batch_size))['loss']) for it in range(14)]
    assert all(np.isfinite(ls)) and (r2._cap is not None or r2.get_cos_anneal_ratio() < 1.0)
    if r2._cap is not None:
        assert np.mean(ls[-4:]) < np.mean(ls[:4]) and len(set(ls[4:])) > 5          # it learns, and the draws differ per replay


def test_geo_runner_trains_from_a_blender_image_set(tmp_path):
    """conf `dataset.data_dir` -> models/nerfset.Dataset (images resident on the device) -> Runner.train_step."""
    from tests.test_datasets import _write_blender_set
    from vqnerf_release_amd.geo.nerf_runner import Runner
    from vqnerf_release_amd.geo.models.nerfset import Dataset
    data = tmp_path / 'scene'
    data.mkdir()
    _write_blender_set(str(data), n=3, H=24, W=32)
    text = open(os.path.join(HERE, 'golden', 'neus_like.conf')).read().replace('./exp/', str(tmp_path) + '/exp/')
    text = text.replace('warm_up_end = 5000', 'warm_up_end = 0').replace('batch_size = 64', 'batch_size = 128')
    import re
    text = re.sub(r'data_dir = [^\n]*', 'data_dir = %s/\n    longint = false' % data, text, count=1)
    torch.manual_seed(0)
    r = Runner(conf_text=text, case='lego')
    assert isinstance(r.dataset, Dataset) and r.dataset.n_images == 3 and r.dataset.images.is_cuda
    r.update_learning_rate()
    losses = [float(r.train_step(r.dataset.gen_random_rays_at(it % 3, r.batch_size))['loss']) for it in range(4)]
    assert all(np.isfinite(losses)) and r.iter_step == 4
    # validation images (nerf_runner.py:236-343): prediction over the input view, alpha, inside-sphere weight, normals
    from PIL import Image
    r.renderer.perturb = 0.0                                     # (the conf's per-ray jitter would make two renders differ)
    imgs = r.validate_image(idx=1, resolution_level=2)
    base = os.path.join(str(tmp_path), 'exp')
    found = {}
    for root, _, files in os.walk(base):
        for f in files:
            if f.endswith('_0_1.png'):
                found[os.path.basename(root)] = np.asarray(Image.open(os.path.join(root, f)))
    assert set(found) == {'validations_fine', 'alpha', 'inside_sphere', 'normals'}
    assert found['validations_fine'].shape == (24, 16, 3) and found['alpha'].shape == (12, 16) and found['normals'].shape == (12, 16, 3)
    np.testing.assert_array_equal(found['validations_fine'][..., ::-1], imgs['validations_fine'])       # file = RGB of the (B,G,R) array
    np.testing.assert_array_equal(found['validations_fine'][12:], r.dataset.image_at(1, 2)[..., ::-1])  # lower half: the input image
    out = r.renderer.render(*[t.reshape(-1, 3).contiguous() for t in r.dataset.gen_rays_at(1, resolution_level=2)],
                            *r.dataset.near_far_from_sphere(torch.zeros(192, 3).cuda(), torch.zeros(192, 3).cuda()), r.dataset.max_radius,
                            cos_anneal_ratio=r.get_cos_anneal_ratio(), background_rgb=torch.ones(1, 3).cuda())
    want = (out['color_fine'].reshape(12, 16, 3) * 256).clip(0, 255).to(torch.uint8).cpu().numpy()
    np.testing.assert_array_equal(imgs['validations_fine'][:12], want)
    bgn = found['normals'][found['alpha'] == 0]
    assert bgn.size == 0 or (np.abs(bgn.astype(int) - int(128 / np.sqrt(3) + 128)) <= 1).all()        # background normal (1,1,1)/sqrt 3


def test_geo_runner_trains_from_a_projection_matrix_set(tmp_path):
    """conf `dataset.data_dir` with train.json (world_mat / scale_mat) -> models/dtuset.Dataset -> Runner.train_step with
    per-ray near / far from the unit sphere (dtu_runner.py:36, dtuset.py:142-149)."""
    from tests.test_datasets import _write_dtu_set
    from vqnerf_release_amd.geo.nerf_runner import Runner
    from vqnerf_release_amd.geo.models.dtuset import Dataset
    data = tmp_path / 'scan'
    data.mkdir()
    _write_dtu_set(data, n=3, H=24, W=32)
    text = open(os.path.join(HERE, 'golden', 'neus_like.conf')).read().replace('./exp/', str(tmp_path) + '/exp/')
    text = text.replace('warm_up_end = 5000', 'warm_up_end = 0').replace('batch_size = 64', 'batch_size = 128')
    import re
    text = re.sub(r'data_dir = [^\n]*', 'data_dir = %s/' % data, text, count=1)
    torch.manual_seed(0)
    r = Runner(conf_text=text, case='scan24')
    assert isinstance(r.dataset, Dataset) and r.dataset.n_images == 3 and r.dataset.images.is_cuda and r.dataset.max_radius == 1.0
    r.update_learning_rate()
    losses = [float(r.train_step(r.dataset.gen_random_rays_at(it % 3, r.batch_size))['loss']) for it in range(4)]
    assert all(np.isfinite(losses)) and r.iter_step == 4


def test_decomp_trainer_trains():
    from oracle import decomp as od
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, specs = od.make_model_params(seed=0, K=15)
    cfg = make_config(n_rays_per_step=256)
    model = load_oracle_params(get_model_class('vq_nfr')(cfg), p, 'cuda')
    pts = od.make_points(512, seed=11)
    batch = make_batch(pts, 'cuda')
    # trainer creates the lazy variables, then the optimiser can see all of them
    model.get_codebook(); _ = model.light
    opt = torch.optim.Adam(model.trainable_variables, lr=5e-4, eps=1e-7, amsgrad=True)
    tr = train_nfr.Trainer(model, opt)
    cb0 = model._codebook.detach().clone()
    with torch.no_grad():
        z0 = model._pred_enc_at(batch[7]).clone()
    losses = []
    for it in range(10):
        wl, _, ld = tr.train_iter(batch, global_bs=512)
        losses.append(float(wl))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert not torch.equal(cb0, model._codebook)                   # EMA moved the codebook
    with torch.no_grad():
        z1 = model._pred_enc_at(batch[7])
        z1_ref = model.net['bottleneck'](model.net['fine_enc'](model.embedder['xyz'](batch[7])))
    assert not torch.equal(z0, z1)
    np.testing.assert_allclose(z1.cpu().numpy(), z1_ref.cpu().numpy(), rtol=0, atol=3e-6)
    # vali / vq_test entry points run on the fused path
    wl, to_vis, ld = train_nfr.vali_iter(model, batch, 512)
    assert np.isfinite(float(wl)) and 'pred_vq_rgb' in to_vis
    ld2 = train_nfr.vali_vq(model, batch)
    assert set(ld2) >= {'rgb', 'vqrgb', 'chromaticity'}


def test_batched_weight_gradients_are_the_per_weight_sequence_bit_for_bit(monkeypatch):
    """vqn_wgrad_finalize (one launch per backward pass: ordered sums of every contraction's partial blocks, written transposed /
    sliced / scaled where they belong) against the per-weight sequence it replaces (vqn_reduce_partials + torch transpose / cat /
    scale): identical gradients, bit for bit, for the reflectance trainer (on the interpreted programs, which keep both tails; the
    dedicated kernels of round 4 only have the batched one) and for the geometry networks."""
    monkeypatch.setenv('VQN_REFL_TRAIN', 'prog')
    from oracle import decomp as od
    from oracle import geo as og
    from vqnerf_release_amd.geo import train_programs as tp
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    from tests.test_gpu_neus_render import _build
    old = tp.BATCHED_WGRAD[0]
    try:
        res = {}
        for batched in (False, True):
            tp.BATCHED_WGRAD[0] = batched
            p, specs = od.make_model_params(seed=0, K=15)
            model = load_oracle_params(get_model_class('vq_nfr')(make_config(n_rays_per_step=256)), p, 'cuda')
            batch = make_batch(od.make_points(300, seed=11), 'cuda')
            model.get_codebook(); _ = model.light
            opt = torch.optim.SGD(model.trainable_variables, lr=0.0)
            tr = train_nfr.Trainer(model, opt)
            with launches() as rec:
                tr.train_iter(batch, global_bs=300)
            assert rec.ran('vqn_wgrad_finalize') == batched and rec.ran('vqn_reduce_partials') == (not batched)
            g_dec = [v.grad.detach().clone() for v in model.trainable_variables if v.grad is not None]
            cfg, sdf, col, var, ren = _build('full')
            o, d, near, far = [torch.tensor(a).cuda() for a in og.make_rays(24, 5)]
            with launches() as rec:
                rr = ren.render(o, d, near, far, 2.0, perturb_overwrite=0, background_rgb=torch.ones(1, 3).cuda(), cos_anneal_ratio=1.0)
                (rr['color_fine'].sum() + rr['gradient_error']).backward()
            assert rec.ran('vqn_wgrad_finalize') == batched and rec.ran('vqn_reduce_partials') == (not batched)
            res[batched] = (g_dec, [q.grad.detach().clone() for m in (sdf, col) for q in m.parameters()])
        for a, b in zip(res[False][0] + res[False][1], res[True][0] + res[True][1]):
            assert a.shape == b.shape and torch.equal(a, b)
        assert len(res[True][0]) > 20 and len(res[True][1]) > 20
    finally:
        tp.BATCHED_WGRAD[0] = old


def test_multi_copy():
    from vqnerf_release_amd import parallel
    g = torch.Generator(device='cuda').manual_seed(0)
    srcs = [torch.randn(n, device='cuda', generator=g) for n in (1, 1023, 1024, 1025, 70000, 3)] + [torch.randn(7, 33, device='cuda', generator=g)]
    srcs += [torch.randn(5, device='cuda', generator=g) for _ in range(120)]                # more than one table
    flat = torch.full((sum(s.numel() for s in srcs) + 4,), float('nan'), device='cuda')
    dsts, o = [], 2
    for s in srcs:
        dsts.append(flat[o:o + s.numel()].view_as(s))
        o += s.numel()
    with launches() as rec:
        parallel.multi_copy(dsts, srcs)
    assert rec.ran('vqn_multi_copy') and rec.counts['vqn_multi_copy'] == 1
    for d, s in zip(dsts, srcs):
        assert torch.equal(d, s)
    assert torch.isnan(flat[:2]).all() and torch.isnan(flat[-2:]).all()
    parallel.multi_copy([], [])


def test_decomp_trainer_graph_with_code_dropout():
    """Trainer(graph=True) with the code-dropout thresholds as a graph input: a code whose threshold is 1 is never assigned
    (its draw in [0, 1) never reaches it), one with threshold 0 always may be; new thresholds take effect at the next replay;
    the draw itself comes from the device generator inside the graph, so consecutive replays differ."""
    from oracle import decomp as od
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, specs = od.make_model_params(seed=0, K=15)
    cfg = make_config(n_rays_per_step=128)
    model = load_oracle_params(get_model_class('vq_nfr')(cfg), p, 'cuda')
    model.get_codebook(); _ = model.light
    opt, _, clip = train_nfr.make_optimizer(cfg, model.trainable_variables, capturable=True)
    tr = train_nfr.Trainer(model, opt, clip=clip, graph=True)
    batch = make_batch(od.make_points(256, seed=3), 'cuda')
    with pytest.raises(ValueError):
        tr.train_iter(batch, global_bs=128, thres=np.zeros(15, np.float32))      # host thresholds cannot be a graph input
    tr._calls = 0
    seen = []
    hook = model.vq_layer.register_forward_hook(lambda m, i, o: seen.append(o['encoding_indices']))
    half = torch.tensor([0.0] * 8 + [1.0] * 7, device='cuda')           # codes 8..14 always dropped
    losses = [float(tr.train_iter(batch, global_bs=128, thres=half)[0]) for _ in range(train_nfr.Trainer.GRAPH_WARMUP + 1)]
    assert tr._captured is not None and all(np.isfinite(losses))
    static_idx = seen[-1]                                                # the captured step's index tensor: refreshed by every replay
    hook.remove()
    assert int(static_idx.max()) <= 7
    mid = torch.tensor([0.5] * 15, device='cuda')                       # every code dropped with probability 1/2, drawn per replay
    used = []
    for _ in range(6):
        l = float(tr.train_iter(batch, global_bs=128, thres=mid)[0])
        assert np.isfinite(l)
        used.append(tuple(sorted(set(static_idx.reshape(-1).tolist()))))
    assert len(set(used)) > 1                                            # different draws -> different code subsets
    only3 = torch.ones(15, device='cuda'); only3[3] = 0.0
    tr.train_iter(batch, global_bs=128, thres=only3)
    assert set(static_idx.reshape(-1).tolist()) == {3}
    with pytest.raises(ValueError):
        tr.train_iter(batch, global_bs=128)                             # recorded with dropout: cannot replay without


def test_decomp_trainer_graph_replays_the_eager_step():
    """Trainer(graph=True): the captured step (forward, loss, backward, EMA codebook move, Adam) replayed on new batches is
    the eager step bit for bit (same kernels, same order; both sides use the capturable Adam so the update arithmetic is
    the same statement)."""
    from oracle import decomp as od
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, specs = od.make_model_params(seed=0, K=15)
    cfg = make_config(n_rays_per_step=128)
    batches = [make_batch(od.make_points(256, seed=20 + i), 'cuda') for i in range(6)]
    runs = {}
    for graph in (False, True):
        model = load_oracle_params(get_model_class('vq_nfr')(cfg), p, 'cuda')
        model.get_codebook(); _ = model.light
        opt, _, clip = train_nfr.make_optimizer(cfg, model.trainable_variables, capturable=True)
        tr = train_nfr.Trainer(model, opt, clip=clip, graph=graph)
        losses = []
        for b in batches:
            wl, to_vis, ld = tr.train_iter(b, global_bs=256)
            losses.append(float(wl))
        assert (tr._captured is not None) == graph
        runs[graph] = (losses, [q.detach().clone() for q in model.trainable_variables], model._codebook.detach().clone(),
                       int(model.vq_layer.ema_dw.counter), to_vis['pred_rgb'].clone())
    (l0, w0, c0, n0, v0), (l1, w1, c1, n1, v1) = runs[False], runs[True]
    assert n0 == n1 == len(batches)
    assert l0 == l1, (l0, l1)
    assert all(torch.equal(a, b) for a, b in zip(w0, w1)) and torch.equal(c0, c1) and torch.equal(v0, v1)
    assert l1[-1] != l1[-2]                                            # the replays really consumed different batches
    # the conditions are checked, not assumed
    with pytest.raises(ValueError):
        tr.train_iter(batches[0], global_bs=256, thres=np.full((1, 15), 0.5, np.float32))
    with pytest.raises(ValueError):
        tr.train_iter(tuple(t[:64] if torch.is_tensor(t) else t[:64] for t in batches[0]), global_bs=256)
    with pytest.raises(ValueError):
        train_nfr.Trainer(model, torch.optim.Adam(model.trainable_variables, lr=1e-3), graph=True)


def test_inference_after_graph_replays_sees_the_moved_weights():
    """ADVICE r02 (medium), and wider than reported: a replayed HIP graph runs Adam and the EMA codebook move without bumping any
    tensor `_version` -- and so does the EAGER fused / capturable Adam (`torch._fused_adam_` leaves `_version` alone, measured).
    Inference-side caches keyed on `_version` alone (weight packs, codebook fragments) would serve STALE weights to every
    validation after the first one.  Train (eager and graph), validate, train more, validate again: each validation must equal
    (a) the other trainer's, bit for bit, and (b) a FRESH model loaded from the trained parameters (no caches at all)."""
    from oracle import decomp as od
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, specs = od.make_model_params(seed=0, K=15)
    cfg = make_config(n_rays_per_step=128, lr=5e-3)
    batches = [make_batch(od.make_points(256, seed=40 + i), 'cuda') for i in range(10)]
    view = make_batch(od.make_points(300, seed=99), 'cuda', bg_every=5)
    keys = ('rgb', 'albedo', 'vq_rgb', 'vq_albedo', 'embed')

    def validate(model):
        with torch.no_grad(), launches() as rec:
            pred, gt, lk, _ = model.call(view, mode='vali')
            emb = model.fast_embed(view, mode='vali')[3]['embed']
            test = model.vq_test(view, mode='vali')[2]
        assert rec.ran('vqn_mlp_chain_vq_fwd')            # the fused front: weight packs AND codebook fragments are cached
        return {k: pred[k].clone() for k in keys} | {'fe': emb.clone(), 'vt': test['vqrgb'].clone()}

    runs = {}
    for graph in (False, True):
        model = load_oracle_params(get_model_class('vq_nfr')(cfg), p, 'cuda')
        model.get_codebook(); _ = model.light
        opt, _, clip = train_nfr.make_optimizer(cfg, model.trainable_variables, capturable=True)
        tr = train_nfr.Trainer(model, opt, clip=clip, graph=graph)
        valis = []
        for i, b in enumerate(batches):
            tr.train_iter(b, global_bs=256)
            if i in (4, 9):                                   # after replays 3..5 and again after replays 6..10
                model.assume_foreground = False               # a validation view has background rows
                got = validate(model)
                fresh = load_oracle_params(get_model_class('vq_nfr')(cfg), p, 'cuda')
                fresh.get_codebook()
                fresh.load_state_dict(model.state_dict())
                want = validate(fresh)
                for k in got:
                    assert torch.equal(got[k], want[k]), (graph, i, k)
                valis.append(got)
                model.assume_foreground = graph
        assert (tr._captured is not None) == graph
        runs[graph] = valis
    for a, b in zip(runs[False], runs[True]):
        for k in a:
            assert torch.equal(a[k], b[k]), k
    # and the two validations differ from each other (the weights did move between them)
    assert not torch.equal(runs[True][0]['albedo'], runs[True][1]['albedo'])
    assert not torch.equal(runs[True][0]['vq_albedo'], runs[True][1]['vq_albedo'])


def test_decomp_trains_from_geometry_buffers_on_disk(tmp_path):
    """The whole hand-off of the pipeline: per-view buffers in the layout gen_geo writes -> datasets.shape_unit (device-resident
    views) -> outer_sample pairs -> Trainer.train_iter (eager, then replayed from the captured HIP graph)."""
    from oracle import decomp as od
    from tests.test_datasets import _write_decomp_view, _decomp_cfg
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.datasets import get_dataset_class
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    rng = np.random.default_rng(3)
    for vid in ('train_000', 'train_001'):
        _write_decomp_view(str(tmp_path / 'data'), str(tmp_path / 'geo'), vid, 24, 32, 512, rng, collapse=False)
    cfg = _decomp_cfg(tmp_path, imh=24, n_rays_per_step=64)
    ds = get_dataset_class('shape_unit')(cfg, 'train', device='cuda')
    p, _ = od.make_model_params(seed=0, K=15)
    model = load_oracle_params(get_model_class('vq_nfr')(cfg), p, 'cuda')
    model.get_codebook(); _ = model.light
    opt, sched, clip = train_nfr.make_optimizer(cfg, model.trainable_variables, capturable=True)
    tr = train_nfr.Trainer(model, opt, clip=clip, sched=sched, graph=True)
    gen = torch.Generator(device='cuda').manual_seed(0)
    losses = []
    for epoch in range(3):
        for view in ds.build_pipeline(seed=epoch):
            batch = train_nfr.outer_sample(view, cfg, 'nerf', generator=gen)
            assert batch[7].shape == (128, 3) and batch[7].is_cuda
            losses.append(float(tr.train_iter(batch, global_bs=64)[0]))
    assert len(losses) == 6 and all(np.isfinite(losses)) and tr._captured is not None


def test_outer_sample_pairs_are_neighbours():
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    H, W = 40, 50
    n = H * W
    g = torch.Generator(device='cuda').manual_seed(0)
    ii, jj = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    xyz = torch.stack([ii, jj, torch.zeros_like(ii)], -1).reshape(n, 3).float().cuda()
    alpha = torch.ones(n, 1).cuda()
    alpha.reshape(H, W)[:, :10] = 0.0                                # a background band
    hw = torch.tensor([[H, W]]).repeat(n, 1).cuda()
    z = torch.zeros(n, 3).cuda()
    batch = (['v'] * n, hw, z, z, z, alpha, alpha.clone(), xyz, z, torch.ones(n, 512).cuda())
    out = train_nfr.outer_sample(batch, make_config(n_rays_per_step=128), 'nerf', generator=g)
    p = out[7]
    assert p.shape == (256, 3) and out[9].shape == (256, 512) and len(out[0]) == 256
    d = (p[0::2, :2] - p[1::2, :2]).abs()
    assert d.max() <= 1 and (d.sum(-1) > 0).all()                  # each pair = a pixel and one of its 8 neighbours
    assert (p[:, 1] >= 10).all() and (out[5] > 0.9).all()          # both foreground


@pytest.mark.parametrize('name,B', [('small', 96), ('full', 48), ('small', 7), ('full', 2), ('full', 700)])
def test_hip_training_programs_match_torch_autograd(name, B, monkeypatch):
    """The explicit forward / backward tile programs (+ compositing backward kernel + weight-gradient contraction)
    against torch autograd over the torch statements of the same modules: loss, every parameter gradient.  B = 700 rays of
    the full networks = 89,600 fine samples = 2,800 point tiles: more than one pass of the persistent workgroups and of the
    split-K weight-gradient partials."""
    from oracle import geo as og
    from tests.test_gpu_neus_render import _build
    # both sides on the SAME sample positions: the no-grad up-sampling passes of the HIP side on the f32-input SDF kernel, as the
    # torch side's (on the step's x3 packs -- the default -- a few samples in 10^5 land in another section of the inverse CDF and their
    # normals differ by more than this test's 5e-4; the default is held to the reference's goldens by test_training_path_grads_vs_reference)
    monkeypatch.setenv('VQN_TRAIN_COARSE', 'f32')
    cfg, sdf, col, var, ren = _build(name)
    if cfg['renderer']['n_importance'] == 0:
        ren.n_importance, ren.up_sample_steps = 16, 4            # exercise the up-sampled path for the small nets too
    o, d, near, far = [torch.tensor(a).cuda() for a in og.make_rays(B, 21)]
    tgt = torch.tensor(np.random.default_rng(4).uniform(0, 1, (B, 3)).astype(np.float32)).cuda()
    mask = (torch.arange(B, device='cuda') % 3 != 0).float()[:, None]
    res = {}
    for backend in ('torch', 'hip'):
        ren.train_backend = backend
        for m in (sdf, col, var):
            m.zero_grad(set_to_none=True)
        with launches() as rec:
            rr = ren.render(o, d, near, far, 2.0, perturb_overwrite=0, background_rgb=torch.ones(1, 3).cuda(), cos_anneal_ratio=0.7)
            loss = ((rr['color_fine'] - tgt) * mask).abs().sum() / mask.sum() + 0.1 * rr['gradient_error'] + \
                0.1 * torch.nn.functional.binary_cross_entropy(rr['weight_sum'].clip(1e-3, 1 - 1e-3), mask)
            loss.backward()
        # 'hip' must be the tile-program engine, not a silent fall-back to the autograd statement it is compared with
        assert ren.last_train_backend == backend
        hip = backend == 'hip'
        assert (rec.ran('vqn_tile_program:prog_fwd') or rec.ran('vqn_neus_train_fwd')) == hip and (rec.ran('vqn_tile_program:prog_sbwd') or rec.ran('vqn_neus_train_bwd')) == hip
        assert rec.ran('vqn_wgrad_partials') == hip and rec.ran('vqn_neus_composite_bwd') == hip
        res[backend] = (loss.item(), {f'{nm}.{k}': p.grad.detach().clone() for nm, m in (('sdf', sdf), ('col', col), ('var', var))
                                      for k, p in m.named_parameters()},
                        {k: v.detach().clone() for k, v in rr.items() if torch.is_tensor(v)})
    lt, gt, rt = res['torch']
    lh, gh, rh = res['hip']
    np.testing.assert_allclose(lh, lt, rtol=2e-5)
    for k in ('color_fine', 'weight_sum', 'weights', 'gradients', 'surf', 'depth', 'cdf_fine', 'inside_sphere', 'weight_max'):
        np.testing.assert_allclose(rh[k].cpu().numpy(), rt[k].cpu().numpy().reshape(rh[k].shape), rtol=0, atol=5e-4, err_msg=k)
    worst = 0.0
    for k, ref in gt.items():
        got = gh[k]
        assert got is not None and got.shape == ref.shape, k
        scale = max(float(ref.abs().max()), 1e-7)
        err = float((got - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 5e-3, (k, err, scale)
    print(f'{name}: loss {lh:.6f} vs {lt:.6f}; worst relative gradient error {worst:.2e}')


@pytest.mark.parametrize('P', [1, 33, 4096 + 17, 40000])
def test_training_forward_on_the_render_kernel_matches_the_interpreted_program(P):
    """vqn_neus_train_fwd (the two-image render kernel leaving the backward's saved tensors) against the interpreted prog_fwd of
    the same engine on the full-size networks: sdf / normals / colours and every saved tensor (valid features of valid points) to
    f32 summation-order differences; odd tile counts leave the second image of the last pair a phantom."""
    from tests.test_gpu_neus_render import _build
    cfg, sdf, col, var, ren = _build('full')
    eng = ren._train_engine(sdf, col)
    assert eng is not None and eng.fused_forward()
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(P)
    x = torch.rand(P, 3, device=dev, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(P, 3, device=dev, generator=g), dim=-1)
    sl = [getattr(sdf, 'lin%d' % l) for l in range(sdf.num_layers - 1)]
    cl = [getattr(col, 'lin%d' % l) for l in range(col.num_layers - 1)]
    with torch.no_grad():
        wbuf, descs, flat = eng.pack([m.effective_weight().float() for m in sl], [m.bias.float() for m in sl],
                                     [m.effective_weight().float() for m in cl], [m.bias.float() for m in cl], want_flat=True)
        Ta, Tb = eng.alloc_tensors(P, dev), eng.alloc_tensors(P, dev)
        for T in (Ta, Tb):
            T['X'].copy_(x)
            T['DIRS'].copy_(d)
            for k in T:
                if k not in ('X', 'DIRS', 'ONES'):
                    T[k].fill_(float('nan'))
        with launches() as rec:
            eng.run('prog_fwd', descs, wbuf, Ta, P)
            eng.run_fused_forward(flat, Tb, P)
        assert rec.ran('vqn_tile_program:prog_fwd') and rec.ran('vqn_neus_train_fwd')
    width = {'E': eng.E, 'OUTF': eng.F, 'EXTR': eng.X, 'SDF': 1, 'N': 3, 'RGB': 3}
    for l in range(eng.nL):
        width['U%d' % (l + 1)] = width['GH%d' % l] = eng.out[l]
    for l in range(eng.nC):
        width['C%d' % (l + 1)] = eng.cout[l]
    for n, w in width.items():
        a, c = Ta[n], Tb[n]
        if a.dim() == 4:
            a = a.permute(0, 3, 1, 2).reshape(a.shape[0] * 32, -1)[:P, :w]
            c = c.permute(0, 3, 1, 2).reshape(c.shape[0] * 32, -1)[:P, :w]
        assert not torch.isnan(c).any(), n
        scale = max(float(a.abs().max()), 1e-6)
        assert float((a - c).abs().max()) <= 5e-6 * scale, (n, float((a - c).abs().max()), scale)
    # the padded feature rows the weight-gradient contraction reads along with the valid ones hold finite numbers
    for n in ('E', 'OUTF', 'EXTR', 'U%d' % eng.skip if eng.skip > 0 else 'U1'):
        nt = (P + 31) // 32
        assert torch.isfinite(Tb[n][:nt - 1]).all(), n
    # the same forward on the exact-split engine (vqn_neus_train_fwd_x3: bf16 piece triples, products to 2^-24): f32-level agreement
    with torch.no_grad():
        Tc = eng.alloc_tensors(P, dev)
        Tc['X'].copy_(x)
        Tc['DIRS'].copy_(d)
        for k in Tc:
            if k not in ('X', 'DIRS', 'ONES'):
                Tc[k].fill_(float('nan'))
        with launches() as rec:
            eng.run_fused_forward_x3([m.effective_weight().float() for m in sl], [m.bias.float() for m in sl],
                                     [m.effective_weight().float() for m in cl], [m.bias.float() for m in cl], Tc, P)
        assert rec.ran('vqn_neus_train_fwd_x3') and rec.ran('vqn_neus_pack_update')
    for n, w in width.items():
        a, c = Ta[n], Tc[n]
        if a.dim() == 4:
            a = a.permute(0, 3, 1, 2).reshape(a.shape[0] * 32, -1)[:P, :w]
            c = c.permute(0, 3, 1, 2).reshape(c.shape[0] * 32, -1)[:P, :w]
        assert torch.isfinite(c).all(), n
        scale = max(float(a.abs().max()), 1e-6)
        assert float((a - c).abs().max()) <= 2e-5 * scale, (n, float((a - c).abs().max()), scale)
    for n in ('E', 'OUTF', 'EXTR', 'U1'):
        assert torch.isfinite(Tc[n][:(P + 31) // 32 - 1]).all(), n


@pytest.mark.parametrize('P', [1, 33, 4096 + 17, 40000])
def test_training_backward_on_the_two_image_engine_matches_the_interpreted_programs(P):
    """vqn_neus_train_bwd (colour backward + tangent pass + reverse sweep in one launch) against the interpreted prog_cbwd +
    prog_sbwd of the same engine, full-size networks, same saved tensors and incoming adjoints: every tensor the weight-gradient
    contraction reads, padding rows and points past P included (zeros on both sides), to f32 summation-order differences."""
    from tests.test_gpu_neus_render import _build
    cfg, sdf, col, var, ren = _build('full')
    eng = ren._train_engine(sdf, col)
    assert eng is not None and eng.fused_backward()
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(P)
    x = torch.rand(P, 3, device=dev, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(P, 3, device=dev, generator=g), dim=-1)
    g_rgb, g_n, g_sdf = (torch.randn(P, k, device=dev, generator=g) for k in (3, 3, 1))
    sl = [getattr(sdf, 'lin%d' % l) for l in range(sdf.num_layers - 1)]
    cl = [getattr(col, 'lin%d' % l) for l in range(col.num_layers - 1)]
    outs = ['DC%d' % l for l in range(eng.nC + 1)] + ['GOUTF', 'ED'] + ['UD%d' % (l + 1) for l in range(eng.nL)] + \
        ['AB%d' % l for l in range(eng.nL)]
    with torch.no_grad():
        wbuf, descs, flat = eng.pack([m.effective_weight().float() for m in sl], [m.bias.float() for m in sl],
                                     [m.effective_weight().float() for m in cl], [m.bias.float() for m in cl], want_flat=True)
        Ta, Tb = eng.alloc_tensors(P, dev), eng.alloc_tensors(P, dev)
        for T in (Ta, Tb):
            T['X'].copy_(x)
            T['DIRS'].copy_(d)
            eng.run_fused_forward(flat, T, P)
            for n in outs:
                T[n].fill_(float('nan'))
        with launches() as rec:
            Ta['DOUT'].copy_(g_rgb * Ta['RGB'] * (1.0 - Ta['RGB']))
            eng.run('prog_cbwd', descs, wbuf, Ta, P)
            Ta['V'].copy_(g_n + Ta['GNCOL'])
            Ta['GS'].copy_(g_sdf)
            eng.run('prog_sbwd', descs, wbuf, Ta, P)
            eng.run_fused_backward(flat, Tb, P, g_rgb, g_n, g_sdf)
        assert rec.ran('vqn_tile_program:prog_sbwd') and rec.ran('vqn_neus_train_bwd')
    nt = (P + 31) // 32

    def goutf_row0(T):
        """round 4: the fused kernels leave d loss / d sdf / scale in row 0 of GOUTF (the interpreter: zeros), so that the contraction
        GOUTF x u_L yields row 0 of the last layer's gradient by itself -- checked here, then taken out of the comparison"""
        want = torch.zeros(nt * 32, device=dev)
        want[:P] = g_sdf.reshape(-1) / eng.scale
        got = T['GOUTF'][:nt, 0, 0, :].reshape(-1)
        assert float((got - want).abs().max()) <= 1e-6 * max(float(want.abs().max()), 1e-6)
        c0 = T['GOUTF'][:nt].clone()
        c0[:, 0, 0, :] = 0.0
        return c0

    for n in outs:
        a, c = Ta[n][:nt], (goutf_row0(Tb) if n == 'GOUTF' else Tb[n][:nt])
        assert torch.isfinite(c).all(), n
        wrote = ~torch.isnan(a)                  # (the interpreter leaves the feature rows past a one-row-quad tensor's eight unwritten)
        assert wrote.float().mean() > 0.2, n
        a, c = a[wrote], c[wrote]
        scale = max(float(a.abs().max()), 1e-6)
        assert float((a - c).abs().max()) <= 5e-6 * scale, (n, float((a - c).abs().max()), scale)
    # the same pass on the exact-split engine (vqn_neus_train_bwd_x3: packs by vqn_pack_x3_gather): f32-level agreement
    with torch.no_grad():
        Tx = eng.alloc_tensors(P, dev)
        Tx['X'].copy_(x)
        Tx['DIRS'].copy_(d)
        eng.run_fused_forward(flat, Tx, P)
        for n in outs:
            Tx[n].fill_(float('nan'))
        with launches() as rec:
            eng.run_fused_backward_x3(flat, Tx, P, g_rgb, g_n, g_sdf)
        assert rec.ran('vqn_neus_train_bwd_x3') and rec.ran('vqn_pack_x3_gather')
    for n in outs:
        a, c = Ta[n][:nt], (goutf_row0(Tx) if n == 'GOUTF' else Tx[n][:nt])
        assert torch.isfinite(c).all(), n
        wrote = ~torch.isnan(a)
        a, c = a[wrote], c[wrote]
        scale = max(float(a.abs().max()), 1e-6)
        assert float((a - c).abs().max()) <= 2e-5 * scale, (n, float((a - c).abs().max()), scale)
    # no incoming adjoint for the normals / the sdf (None in autograd): the same as zeros
    with torch.no_grad():
        Tc = eng.alloc_tensors(P, dev)
        Tc['X'].copy_(x)
        Tc['DIRS'].copy_(d)
        eng.run_fused_forward(flat, Tc, P)
        eng.run_fused_backward(flat, Tb, P, g_rgb, torch.zeros_like(g_n), torch.zeros_like(g_sdf))
        eng.run_fused_backward(flat, Tc, P, g_rgb, None, None)
    for n in outs:
        assert torch.equal(Tb[n][:nt], Tc[n][:nt]), n


@pytest.mark.parametrize('mode', ['prog', 'fused', 'x3'])
def test_training_forward_switch(mode, monkeypatch):
    """VQN_TRAIN_FWD / VQN_TRAIN_BWD select the forward / backward of the training engine (the two-image kernels or the interpreted
    programs); both reach the same gradients (to f32 rounding)."""
    from oracle import geo as og
    from tests.test_gpu_neus_render import _build
    monkeypatch.setenv('VQN_TRAIN_FWD', mode)
    monkeypatch.setenv('VQN_TRAIN_BWD', mode)
    cfg, sdf, col, var, ren = _build('full')
    B = 24
    o, d, near, far = [torch.tensor(a).cuda() for a in og.make_rays(B, 5)]
    with launches() as rec:
        rr = ren.render(o, d, near, far, 2.0, perturb_overwrite=0, background_rgb=torch.ones(1, 3).cuda(), cos_anneal_ratio=1.0)
        (rr['color_fine'].sum() + rr['gradient_error']).backward()
    assert ('vqn_neus_train_fwd' in rec.names) == (mode == 'fused') and ('vqn_neus_train_fwd_x3' in rec.names) == (mode == 'x3')
    assert rec.ran('vqn_tile_program:prog_fwd') == (mode == 'prog')
    assert ('vqn_neus_train_bwd' in rec.names) == (mode == 'fused') and ('vqn_neus_train_bwd_x3' in rec.names) == (mode == 'x3')
    assert rec.ran('vqn_tile_program:prog_sbwd') == (mode == 'prog')
    # the up-sampling passes of a training render read the step's x3 packs when the forward runs on the exact-split engine
    assert ('vqn_neus_sdf_points_x3' in rec.names) == (mode == 'x3') and ('vqn_neus_sdf_points' in rec.names) == (mode != 'x3')
    grads = torch.cat([p.grad.reshape(-1) for m in (sdf, col) for p in m.parameters()])
    assert torch.isfinite(grads).all()
    test_training_forward_switch.seen = getattr(test_training_forward_switch, 'seen', {})
    test_training_forward_switch.seen[mode] = grads
    seen = test_training_forward_switch.seen
    for other in seen:
        if other != mode:                      # (x3: its own up-sampling arithmetic moves a few sample positions -- looser bound)
            tol = 2e-3 if 'x3' in (other, mode) else 2e-4
            assert float((seen[other] - grads).abs().max()) <= tol * float(grads.abs().max()), (other, mode)


def test_empty_and_degenerate_inputs():
    """N = 0 is a no-op for every entry point of the training engine; a 1-point tile program runs."""
    import ctypes
    from vqnerf_release_amd import _C
    lib = _C.lib()
    z = ctypes.c_void_p(0)
    one = torch.zeros(64, device='cuda')
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    assert lib.vqn_neus_composite_bwd(p(one), p(one), p(one), p(one), p(one), p(one), p(one), p(one), z, ctypes.c_int64(0), 8,
                                      ctypes.c_float(2.0), ctypes.c_float(1.0), p(one), z, z, z, z, p(one), p(one), p(one), p(one), z) == 0
    assert lib.vqn_brdf_shade_fwd(z, z, z, z, z, z, z, ctypes.c_int64(0), 512, 1, z, z, z, z, z, z, z, z, z, z, z, z, 0, z, 0, z, z) == 0
    assert lib.vqn_mlp_chain_fwd(p(one), p(one), z, ctypes.c_int64(0), z, 0, z, 0, z, 0, z, 0, z) == 0
    rc = lib.vqn_wgrad_partials(p(one), 1, 0, 9, p(one), 1, 0, 1, ctypes.c_int64(1), 4, p(one), z)
    assert rc == -2 and b'feature tiles' in lib.vqn_last_error()
    rc = lib.vqn_brdf_shade_fwd(p(one), p(one), p(one), z, p(one), p(one), p(one), ctypes.c_int64(4), 100, 1, p(one), p(one), p(one),
                                z, z, z, z, z, p(one), z, z, z, 0, z, 0, z, z)
    assert rc == -2 and b'256, 512 or 1024' in lib.vqn_last_error()


@pytest.mark.parametrize('N,F,tiles', [(100, 70, None), (1, 3, 1), (4097, 256, 8), (64, 33, 4)])
def test_tfmt_pack_and_unpack(N, F, tiles):
    """vqn_tfmt_pack / vqn_tfmt_unpack against the index formula of include/vqn_vm_desc.h: element (p, f) of the rows sits at
    [p // 32][f // 32][f % 32][p % 32]; padding rows / features are zero; strided inputs are accepted."""
    from vqnerf_release_amd.decomp.train_programs import to_tfmt, from_tfmt
    g = torch.Generator(device='cuda').manual_seed(N + F)
    wide = torch.rand((N, F + 5), device='cuda', generator=g)
    x = wide[:, :F]                                                  # row stride F + 5
    t = to_tfmt(x, tiles)
    nt, ft = (N + 31) // 32, tiles or (F + 31) // 32
    ref = torch.zeros(nt * 32, ft * 32, device='cuda')
    ref[:N, :F] = x
    assert torch.equal(t, ref.view(nt, 32, ft, 32).permute(0, 2, 3, 1).contiguous())
    assert torch.equal(from_tfmt(t, N, F), x)
    out = torch.full((nt, ft, 32, 32), 7.0, device='cuda')
    assert to_tfmt(x.contiguous(), tiles, out=out) is out and torch.equal(out, t)


def test_fused_weight_norm_matches_torch_autograd():
    """vqn_weight_norm_fwd / _bwd for a list of layers against nn.utils.weight_norm's expression under torch autograd."""
    from vqnerf_release_amd.geo.models.fields import _Lin, effective_weights
    torch.manual_seed(0)
    lins = [_Lin(39, 256, True), _Lin(256, 217, True), _Lin(7, 3, True), _Lin(256, 257, False), _Lin(295, 256, True)]
    lins = [m.cuda() for m in lins]
    with torch.no_grad():
        for m in lins:
            if m.weight_norm:
                m.weight_g.mul_(1.3)
    ws = effective_weights(lins)
    ref = [m.effective_weight() for m in lins]
    for w, r in zip(ws, ref):
        np.testing.assert_allclose(w.detach().cpu().numpy(), r.detach().cpu().numpy(), rtol=5e-7, atol=0)   # row norms summed in another order: <= 2 ulp
    seeds = [torch.randn_like(w) for w in ws]
    sum((w * s).sum() for w, s in zip(ws, seeds)).backward()
    got = [(m.weight_g.grad.clone(), m.weight_v.grad.clone()) if m.weight_norm else (m.weight.grad.clone(),) for m in lins]
    for m in lins:
        m.zero_grad(set_to_none=True)
    sum((w * s).sum() for w, s in zip(ref, seeds)).backward()
    for m, g in zip(lins, got):
        want = (m.weight_g.grad, m.weight_v.grad) if m.weight_norm else (m.weight.grad,)
        for a, b in zip(g, want):
            scale = float(b.abs().max())
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=0, atol=2e-6 * scale)


@pytest.mark.parametrize('a_nt,b_nt,n_tiles', [(8, 8, 300), (8, 3, 77), (5, 8, 40), (8, 8, 1)])
def test_bf16x3_weight_gradient_contraction_matches_the_f32_one(a_nt, b_nt, n_tiles):
    """vqn_wgrad_partials_x3 (every operand split exactly into three bf16 pieces, six MFMAs per product) against
    vqn_wgrad_partials and against the float64 contraction: f32-level agreement over eight decades of operand scale (no range
    caveat, unlike an f16 split), bias-gradient row sums identical."""
    import ctypes
    from vqnerf_release_amd import _C
    lib = _C.lib()
    g = torch.Generator(device='cuda'); g.manual_seed(a_nt * 100 + b_nt)
    A = torch.randn((n_tiles, a_nt, 32, 32), device='cuda', generator=g) * torch.exp(torch.randn((n_tiles, 1, 1, 32), device='cuda', generator=g) * 4.0) * 1e-6
    B = torch.randn((n_tiles, b_nt, 32, 32), device='cuda', generator=g)
    n_split = 64
    outs = {}
    for entry in ('vqn_wgrad_partials', 'vqn_wgrad_partials_x3'):
        ws = torch.zeros((n_split, a_nt * 32, b_nt * 32), device='cuda')
        rs = torch.zeros((n_split, a_nt * 32), device='cuda')
        n = getattr(lib, entry)(_C._ptr(A), a_nt, 0, a_nt, _C._ptr(B), b_nt, 0, b_nt, ctypes.c_int64(n_tiles), n_split, _C._ptr(ws),
                                _C._ptr(rs), _C._stream())
        assert n > 0, lib.vqn_last_error()
        torch.cuda.synchronize()
        outs[entry] = (ws[:n].double().sum(0), rs[:n].double().sum(0))
    Ad, Bd = A.double().permute(1, 2, 0, 3).reshape(a_nt * 32, -1), B.double().permute(1, 2, 0, 3).reshape(b_nt * 32, -1)
    want = Ad @ Bd.T
    scale = (Ad.abs() @ Bd.abs().T).max()                     # the sum of |terms|: what fp32 rounding is relative to
    e32 = float((outs['vqn_wgrad_partials'][0] - want).abs().max() / scale)
    ex3 = float((outs['vqn_wgrad_partials_x3'][0] - want).abs().max() / scale)
    assert ex3 <= max(2.0 * e32, 2e-7), (ex3, e32)
    assert torch.equal(outs['vqn_wgrad_partials'][1], outs['vqn_wgrad_partials_x3'][1]) or \
        float((outs['vqn_wgrad_partials'][1] - outs['vqn_wgrad_partials_x3'][1]).abs().max()) <= 1e-6 * float(Ad.abs().sum(1).max())


@pytest.mark.parametrize('N,F,act', [(100, 3, 3), (33, 1, 3), (4097, 256, 1), (64, 70, 0), (5, 3, 3)])
def test_pack_delta_is_unpack_times_activation_derivative_pack(N, F, act):
    """vqn_tfmt_pack_delta (the top-layer delta of a backward program in one launch) against the sequence it replaces: unpack the
    saved output, g * y * (1 - y) (sigmoid) / g * (y > 0) (ReLU) / g, pack -- bit for bit, zero padding included; g = None: zeros."""
    from vqnerf_release_amd.decomp.train_programs import to_tfmt, from_tfmt, pack_delta
    g = torch.Generator(device='cuda').manual_seed(N + F)
    y_rows = torch.rand(N, F, device='cuda', generator=g) - (0.5 if act == 1 else 0.0)
    gr = torch.randn(N, F, device='cuda', generator=g)
    Y = to_tfmt(y_rows)
    y = from_tfmt(Y, N, F)
    ref = to_tfmt(gr * y * (1 - y) if act == 3 else (gr * (y > 0) if act == 1 else gr), Y.shape[1])
    out = torch.full_like(Y, float('nan'))
    pack_delta(gr, Y, act, N, F, out)
    assert torch.equal(out, ref)
    pack_delta(None, Y, act, N, F, out)
    assert torch.equal(out, torch.zeros_like(out))


def test_prepared_step_packs_are_not_reused_after_the_weights_moved():
    """NeusTrainEngine.prepare_step packs the step's weights for the up-sampling passes and the forward; a forward that runs later on
    OTHER weights (an in-place edit, an optimiser step in between) must re-pack: keyed on the weights epoch + parameter versions."""
    from oracle import geo as og
    from tests.test_gpu_neus_render import _build
    cfg, sdf, col, var, ren = _build('full')
    o, d, near, far = [torch.tensor(a).cuda() for a in og.make_rays(8, 3)]
    eng = ren._train_engine(sdf, col)
    assert eng.forward_mode() == 'x3'
    s_l = [getattr(sdf, 'lin%d' % l) for l in range(sdf.num_layers - 1)]
    c_l = [getattr(col, 'lin%d' % l) for l in range(col.num_layers - 1)]
    assert eng.prepare_step(s_l, c_l) is not None and eng._x3_prepared is not None
    with torch.no_grad():
        sdf.lin8.bias.add_(0.05)                                  # the packs made above are stale now
    z = torch.linspace(0.0, 1.0, 16, device='cuda')[None, :] * (far - near) + near
    with launches() as rec:
        rc = ren.render_core(o, d, z.contiguous(), 2.0 / 16, 2.0, sdf, var, col, background_rgb=torch.ones(1, 3).cuda(), cos_anneal_ratio=1.0)
    assert ren.last_train_backend == 'hip' and rec.counts.get('vqn_neus_pack_update', 0) == 1      # re-packed, not reused
    with torch.no_grad():
        ref = sdf.sdf(((o[:, None, :] + d[:, None, :] * rc['mid_z_vals'][..., None]).reshape(-1, 3)).contiguous())
    assert float((rc['sdf'].detach().reshape(-1) - ref.reshape(-1)).abs().max()) < 2e-5


@pytest.mark.parametrize('amsgrad,tensor_lr,wd', [(True, True, 0.0), (False, True, 0.0), (True, False, 0.01), (False, False, 0.0)])
def test_hip_adam_is_torch_adam(amsgrad, tensor_lr, wd):
    """optim.HipAdam (vqn_adam_step, one launch) against torch.optim.Adam(capturable=True, fused=True): parameters and the whole
    optimiser state after 6 steps on the same gradients, state_dict interchangeable."""
    from vqnerf_release_amd.optim import HipAdam
    g = torch.Generator(device='cuda').manual_seed(0)
    shapes = [(7,), (1,), (), (300, 257), (1025,), (64, 64)] + [(33,)] * 60          # more tensors than one kernel table holds
    mk = lambda: [torch.nn.Parameter(t.clone()) for t in init]
    init = [torch.randn(s, device='cuda', generator=g) for s in shapes]
    pa, pb = mk(), mk()
    lr = lambda: torch.tensor(3e-3, device='cuda') if tensor_lr else 3e-3
    oa = torch.optim.Adam(pa, lr=lr(), eps=1e-7, amsgrad=amsgrad, weight_decay=wd, capturable=True, fused=True)
    ob = HipAdam(pb, lr=lr(), eps=1e-7, amsgrad=amsgrad, weight_decay=wd)
    for it in range(6):
        grads = [torch.randn(s, device='cuda', generator=g) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if it == 3:
            pa[0].grad = pb[0].grad = None                    # a parameter without a gradient is skipped (its step count too)
        with launches() as rec:
            oa.step()
            ob.step()
        assert rec.counts['vqn_adam_step'] == 1
    for p, q in zip(pa, pb):
        assert float((p - q).detach().abs().max()) <= 2e-6 * max(1.0, float(p.detach().abs().max()))
    sa, sb = oa.state_dict()['state'], ob.state_dict()['state']
    assert sa.keys() == sb.keys()
    for k in sa:
        assert sa[k].keys() == sb[k].keys()
        for name in sa[k]:
            a, b = sa[k][name].float(), sb[k][name].float()
            assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), (k, name)
    ob.load_state_dict(oa.state_dict())                          # interchangeable checkpoints


def test_pack_gather_leaves_the_f32_images_in_the_same_launch():
    """vqn_pack_x3_gather2 (round 4): the piece pack of vqn_pack_x3_gather and, from the same flat vector, flat[fidx] -- one launch."""
    from vqnerf_release_amd import _C
    g = torch.Generator(device='cuda').manual_seed(3)
    flat = torch.randn(50000, device='cuda', generator=g)
    gidx = torch.randint(0, 50000, (37 * 512,), device='cuda', generator=g, dtype=torch.int32)
    fidx = torch.randint(0, 50000, (1234,), device='cuda', generator=g, dtype=torch.int32)
    want = _C.pack_x3_gather(flat, gidx, 37)
    got, wf = _C.pack_x3_gather(flat, gidx, 37, fidx)
    assert torch.equal(got, want) and torch.equal(wf, flat[fidx.long()])
    with pytest.raises(_C.VqnError):
        _C.pack_x3_gather(flat, gidx, 37, fidx.long())


def test_commitment_cost_inside_the_quantiser_kernels_rounds_like_the_framework():
    """loss_post of vqn_vq_quantize_rows_train / vqn_vq_train_bwd (round 4): (mean) * cost and g * cost as their own f32 multiplications --
    what `commitment_cost * e_latent_loss` and its autograd did with two framework launches on one number."""
    from vqnerf_release_amd import _C
    g = torch.Generator(device='cuda').manual_seed(4)
    N, D, K = 3001, 256, 15
    z = torch.rand(N, D, device='cuda', generator=g)
    cb = torch.nn.functional.normalize(torch.rand(D, K, device='cuda', generator=g), dim=0).contiguous()
    cost = 0.1
    idx1, ste1, loss1, cnt1, xn1 = _C.vq_quantize_rows(z, cb, want_ste=True, want_xnorm=True)
    idx2, ste2, loss2, cnt2, xn2 = _C.vq_quantize_rows(z, cb, want_ste=True, want_xnorm=True, loss_post=cost)
    assert torch.equal(idx1, idx2) and torch.equal(ste1, ste2) and torch.equal(xn1, xn2) and torch.equal(cnt1, cnt2)
    assert torch.equal(loss2, cost * loss1)
    gs = torch.randn(N, D, device='cuda', generator=g)
    gl = torch.tensor(0.7, device='cuda')
    a = _C.vq_train_bwd(z, xn1, ste1, gs, gl * cost)
    b = _C.vq_train_bwd(z, xn1, ste1, gs, gl, loss_post=cost)
    assert torch.equal(a, b)


@pytest.mark.parametrize('eps_mode,amsgrad', [('keras', True), ('keras', False), ('torch', True)])
def test_hip_adam_against_the_oracle_statements(eps_mode, amsgrad):
    """vqn_adam_step against oracle/optim.py: Keras Adam(amsgrad) -- epsilon on the UN-debiased sqrt(vhat), the
    reflectance trainer's optimiser (train_nfr.py:127-138) -- and torch.optim.Adam, three steps whose gradients span 1e-9 .. 1 so
    that the epsilon placement is what is being tested (at |g| ~ eps the two updates differ by up to 16 x at t = 1)."""
    from oracle import optim as oo
    from vqnerf_release_amd.optim import HipAdam
    rng = np.random.default_rng(5)
    shapes = [(1000,), (64, 33), (5,)]
    init = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    ps = [torch.nn.Parameter(torch.tensor(a, device='cuda')) for a in init]
    lr = 2e-3
    opt = HipAdam(ps, lr=torch.tensor(lr, device='cuda'), eps=1e-7, amsgrad=amsgrad, eps_mode=eps_mode)
    st = [[a.copy(), 0 * a, 0 * a, 0 * a] for a in init]
    step = oo.keras_adam_step if eps_mode == 'keras' else oo.torch_adam_step
    for t in range(1, 4):
        for k, (p, s) in enumerate(zip(ps, shapes)):
            g = (rng.standard_normal(s) * 10.0 ** rng.integers(-9, 1, size=s)).astype(np.float32)
            p.grad = torch.tensor(g, device='cuda')
            st[k] = list(step(*st[k][:1], g, *st[k][1:], t, lr, eps=1e-7, amsgrad=amsgrad))
        with launches() as rec:
            opt.step()
        assert rec.counts['vqn_adam_step'] == 1
        for k, p in enumerate(ps):
            got, want = p.detach().cpu().numpy(), st[k][0]
            assert np.abs(got - want).max() <= 2e-7 * max(1.0, np.abs(want).max()) + lr * 2e-6, (eps_mode, t, k)
            assert np.allclose(opt.state[p]['exp_avg_sq'].cpu().numpy(), st[k][2], rtol=3e-6, atol=1e-30)
    if eps_mode == 'keras':                                  # and the placement is visible: the torch-placement update is NOT the same
        other = [torch.nn.Parameter(torch.tensor(a, device='cuda')) for a in init]
        o2 = HipAdam(other, lr=lr, eps=1e-7, amsgrad=amsgrad, eps_mode='torch')
        g = torch.full((1000,), 1e-7, device='cuda')
        other[0].grad = g
        o2.step()
        mine = [torch.nn.Parameter(torch.tensor(a, device='cuda')) for a in init]
        o3 = HipAdam(mine, lr=lr, eps=1e-7, amsgrad=amsgrad, eps_mode='keras')
        mine[0].grad = g
        o3.step()
        d_t = float((other[0].detach().cpu() - torch.tensor(init[0])).abs().mean())
        d_k = float((mine[0].detach().cpu() - torch.tensor(init[0])).abs().mean())
        assert 0.05 < d_k / d_t < 0.07, (d_k, d_t)           # lr g / (|g| + eps / sqrt(1 - b2)) vs lr g / (|g| + eps) at |g| = eps: 0.0613


@pytest.mark.parametrize('mode', ['x3', 'f32'])
def test_training_backward_is_linear_in_the_adjoints_at_a_large_point_count(mode):
    """A size-independent property of the backward kernels at a point count the interpreter comparison does not reach (65,537 points =
    2,049 point tiles: more than eight passes of the persistent workgroups, an odd tile count): every output of vqn_neus_train_bwd(_x3)
    is linear in the incoming adjoints (d rgb, d n, d sdf) -- bwd(a g1 + b g2) = a bwd(g1) + b bwd(g2) -- and zero for zero adjoints."""
    from tests.test_gpu_neus_render import _build
    cfg, sdf, col, var, ren = _build('full')
    eng = ren._train_engine(sdf, col)
    dev, P = torch.device('cuda'), 65537
    g = torch.Generator(device='cuda').manual_seed(7)
    x = torch.rand(P, 3, device=dev, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(P, 3, device=dev, generator=g), dim=-1)
    adj = [[torch.randn(P, k, device=dev, generator=g) for k in (3, 3, 1)] for _ in range(2)]
    a, b = 0.75, -1.5
    sl = [getattr(sdf, 'lin%d' % l) for l in range(sdf.num_layers - 1)]
    cl = [getattr(col, 'lin%d' % l) for l in range(col.num_layers - 1)]
    outs = ['DC%d' % l for l in range(eng.nC + 1)] + ['GOUTF', 'ED'] + ['UD%d' % (l + 1) for l in range(eng.nL)] + \
        ['AB%d' % l for l in range(eng.nL)]
    run = eng.run_fused_backward_x3 if mode == 'x3' else eng.run_fused_backward
    with torch.no_grad():
        wbuf, descs, flat = eng.pack([m.effective_weight().float() for m in sl], [m.bias.float() for m in sl],
                                     [m.effective_weight().float() for m in cl], [m.bias.float() for m in cl], want_flat=True)
        T = eng.alloc_tensors(P, dev)
        T['X'].copy_(x)
        T['DIRS'].copy_(d)
        eng.run_fused_forward(flat, T, P)
        res = []
        for gr in (adj[0], adj[1], [a * u + b * v for u, v in zip(adj[0], adj[1])], [torch.zeros_like(u) for u in adj[0]]):
            for n in outs:
                T[n].fill_(float('nan'))
            run(flat, T, P, gr[0].contiguous(), gr[1].contiguous(), gr[2].contiguous())
            res.append({n: T[n].clone() for n in outs})
    for n in outs:
        lin = a * res[0][n] + b * res[1][n]
        scale = max(float(lin.abs().max()), 1e-6)
        assert torch.isfinite(res[2][n]).all(), n
        assert float((res[2][n] - lin).abs().max()) <= 2e-5 * scale, (n, float((res[2][n] - lin).abs().max()), scale)
        assert float(res[3][n].abs().max()) == 0.0, n


@pytest.mark.parametrize('mode', ['x3', 'f32'])
def test_training_forward_outputs_are_the_render_kernels(mode):
    """vqn_neus_train_fwd(_x3) is the fine render kernel with extra stores: at 65,537 points its sdf / normals / colours equal
    vqn_neus_fine_points(_x3) on the networks' own packs (the training packs -- one gather from the flat vector, or the
    library's builder -- are the render packs), and the saved U_8 reproduces the sdf through the last layer."""
    from vqnerf_release_amd import _C
    from tests.test_gpu_neus_render import _build
    cfg, sdf, col, var, ren = _build('full')
    eng = ren._train_engine(sdf, col)
    dev, P = torch.device('cuda'), 65537
    g = torch.Generator(device='cuda').manual_seed(11)
    x = torch.rand(P, 3, device=dev, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(P, 3, device=dev, generator=g), dim=-1)
    sl = [getattr(sdf, 'lin%d' % l) for l in range(sdf.num_layers - 1)]
    cl = [getattr(col, 'lin%d' % l) for l in range(col.num_layers - 1)]
    with torch.no_grad():
        W, b = [m.effective_weight().float() for m in sl], [m.bias.float() for m in sl]
        Wc, bc = [m.effective_weight().float() for m in cl], [m.bias.float() for m in cl]
        T = eng.alloc_tensors(P, dev)
        T['X'].copy_(x)
        T['DIRS'].copy_(d)
        if mode == 'x3':
            eng.run_fused_forward_x3(W, b, Wc, bc, T, P)
        else:
            eng.run_fused_forward(eng.pack(W, b, Wc, bc, want_flat=True)[2], T, P)
        m = 'x3' if mode == 'x3' else 'f32'
        wb_s, d_s = sdf.packs(max_tiles=col.max_tiles(), mode=m)
        wb_c, d_c = col.packs(feat_tiles=sdf.plan(mode=m).tiles[-1], mode=m)
        s_ref, n_ref, c_ref = _C.neus_fine_points(d_s, wb_s, d_c, wb_c, pts=x, dirs=d, mode=m)
    # (the sdf bit for bit; normals and colours to an ulp-level bound: the two instantiations of the kernel template may contract
    #  w * (1 - e) of the reverse sweep's first step into an fma differently, since one of them also stores the product)
    assert torch.equal(T['SDF'].reshape(-1), s_ref)
    assert float((T['N'] - n_ref).abs().max()) <= 5e-6 * float(n_ref.abs().max()) and float((T['RGB'] - c_ref).abs().max()) <= 5e-6
    # OUTF row 0 is the raw sdf output: sdf * scale
    outf0 = T['OUTF'].permute(0, 3, 1, 2).reshape(-1, T['OUTF'].shape[1] * 32)[:P, 0]
    assert float((outf0 / float(sdf.scale) - s_ref).abs().max()) <= 1e-6 * max(1.0, float(s_ref.abs().max()))


@pytest.mark.parametrize('model_name,data_type', [('vq_nfr', 'nerf'), ('vq_nfr', 'dtu'), ('vq_nfr', 'hw'), ('nfr_unit', 'nerf'), ('nfr_unit', 'dtu'),
                                                  ('ref_nfr', 'nerf'), ('ref_nfr', 'hw')])
def test_captured_step_is_the_eager_step_for_every_stage_and_data_type(model_name, data_type):
    """Round 5: `train_nfr.fit` / `fit_stage` replay the captured step BY DEFAULT, so every model they can be handed must record: the three
    stages x the data types with (`nerf`) and without (`dtu`, `hw`: learnable display curve) light-visibility rows.  Eight steps of
    Trainer(graph=True) (two eager warm-up steps, the capture, five replays) against eight eager steps from the same state: parameters bit
    for bit."""
    from oracle import decomp as od
    from tests.decomp_util import make_config, make_batch
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    cfg = make_config(model=model_name, data_type=data_type, n_rays_per_step=128, lr=2e-3)
    nerf = data_type == 'nerf'

    def build():
        m = get_model_class(model_name)(cfg)
        m.build_nets(device='cuda', seed=4).to('cuda')
        if model_name == 'vq_nfr':
            cb = np.random.default_rng(0).uniform(0, 1, (15, 256)).astype(np.float32)
            m.set_codebook(cb / np.linalg.norm(cb, axis=1, keepdims=True))
        if model_name == 'ref_nfr':
            for name in ('fine_enc', 'bottleneck', 'spec_out'):
                for p in m.net[name].parameters():
                    p.requires_grad_(False)
            m.set_light(np.full((16, 32, 3), 0.5, np.float32))
        _ = m.light
        if not nerf:
            _ = m.gamma
        m.register_trainable()
        return m
    batches = []
    for i in range(8):
        pts = od.make_points(256, seed=70 + i, lvis=nerf)
        b = make_batch(pts, 'cuda')
        if model_name == 'ref_nfr':
            ref = torch.tensor(np.random.default_rng(90 + i).uniform(0, 1, (256, 3)).astype(np.float32)).cuda()
            b = b[:9] + (ref,) + b[9:]
        batches.append(b)
    finals = {}
    for graph in (False, True):
        m = build()
        opt, _, clip = train_nfr.make_optimizer(cfg, m.trainable_variables, capturable=True)
        tr = train_nfr.Trainer(m, opt, clip=clip, graph=graph)
        for b in batches:
            tr.train_iter(b, global_bs=128)
        torch.cuda.synchronize()
        assert (tr._captured is not None) == graph
        finals[graph] = {k: v.detach().clone() for k, v in m.state_dict().items()}
    assert finals[True].keys() == finals[False].keys()
    moved = 0
    init = {k: v.detach().clone() for k, v in build().state_dict().items()}
    for k in finals[True]:
        assert torch.equal(finals[True][k], finals[False][k]), k
        moved += int(k in init and not torch.equal(finals[True][k], init[k]))
    assert moved >= 6
