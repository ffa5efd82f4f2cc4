"""GPU, world_size 2 over gloo on ONE card: the geo trainer's data-parallel step (HIP kernels + one flat bucket all-reduce) on
two half batches equals the single-process step on the whole batch -- the loss is normalised by GLOBAL sums and the
gradients are summed, so parameters after the step agree to fp32 rounding, and the two replicas stay bit-identical."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _runner(tmp):
    from vqnerf_release_amd.geo.nerf_runner import Runner, SyntheticDataset
    text = open(os.path.join(HERE, 'golden', 'neus_like.conf')).read().replace('./exp/', tmp + '/exp/')
    text = text.replace('warm_up_end = 5000', 'warm_up_end = 0')
    torch.manual_seed(0)
    r = Runner(conf_text=text, case='dp', dataset=SyntheticDataset(n_images=2, H=32, W=32, seed=3))
    r.renderer.perturb = 0.0
    r.update_learning_rate()
    return r


def _batch(r, n=192):
    r.dataset.gen.manual_seed(11)
    return r.dataset.gen_random_rays_at(0, n)


def _worker(rank, world, port, tmp, q):
    import torch.distributed as dist
    from vqnerf_release_amd import parallel
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        r = _runner(tmp)
        data = _batch(r)
        lo, hi = parallel.shard_range(data.shape[0])
        st = r.train_step(data[lo:hi].contiguous())
        grads = r.bucket.flat[:r.bucket.n_grad].detach().cpu().numpy().copy()          # summed over the two ranks
        st = r.train_step(data[lo:hi].contiguous())
        parallel.assert_replicas_identical(list(r.sdf_network.parameters()) + list(r.color_network.parameters()))
        if rank == 0:
            q.put(dict(loss=float(st['loss']), grads=grads))
    finally:
        dist.destroy_process_group()


def test_geo_trainer_dp2_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rk, 2, port, str(tmp_path), q)) for rk in range(2)]
    for p in procs:
        p.start()
    try:
        import time
        t0 = time.time()
        while q.empty():                                           # (the result is MBs: take it before joining, or rank 0 blocks in put)
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), 'a rank died'
            assert time.time() - t0 < 240, 'ranks did not finish'
            time.sleep(0.2)
        got = q.get()
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:                                            # never leave a rank behind on the card
            if p.is_alive():
                p.kill()
    r = _runner(str(tmp_path))
    data = _batch(r)
    st = r.train_step(data)
    want = r.bucket.flat[:r.bucket.n_grad].detach().cpu().numpy().copy()
    st = r.train_step(data)
    # the second step's loss has seen one Adam update on each side: Adam turns rounding-level gradient differences into
    # lr-sized parameter differences where a gradient is ~0, hence the looser bound on it
    assert abs(got['loss'] - float(st['loss'])) <= 2e-3 * max(1.0, abs(float(st['loss'])))
    err = float(np.abs(got['grads'] - want).max()) / float(np.abs(want).max())
    # same gradient up to the order of the sums and the eikonal normaliser (a rank averages over ITS in-sphere samples and the
    # ranks are combined by their share of rays: a weighted mean of ratios, not the ratio of global sums)
    assert err <= 2e-4, err


def _graph_worker(rank, world, port, tmp, q):
    """Runner(graph=True) under data parallelism: the captured step is cut at its collectives (loss normalisers, gradient bucket)
    into graph segments (parallel.SegmentedCapture); replayed steps must equal the eager DP steps of a twin runner bit for bit."""
    import torch.distributed as dist
    from vqnerf_release_amd import parallel
    from vqnerf_release_amd.geo.nerf_runner import Runner, SyntheticDataset
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        text = open(os.path.join(HERE, 'golden', 'neus_like.conf')).read().replace('./exp/', tmp + '/exp/')
        text = text.replace('warm_up_end = 5000', 'warm_up_end = 0')
        out = {}
        for graph in (False, True):
            torch.manual_seed(0)
            r = Runner(conf_text=text, case='dpg', dataset=SyntheticDataset(n_images=2, H=32, W=32, seed=3), graph=True)
            r.graph = graph                                  # same capturable Adam on both sides
            r.renderer.perturb = 0.0
            r.update_learning_rate()
            r.dataset.gen.manual_seed(11)
            losses = []
            for it in range(6):
                data = r.dataset.gen_random_rays_at(it % 2, 192)
                lo, hi = parallel.shard_range(data.shape[0])
                losses.append(float(r.train_step(data[lo:hi].contiguous())['loss']))
            parallel.assert_replicas_identical(list(r.sdf_network.parameters()) + list(r.color_network.parameters()))
            out[graph] = (losses, [p.detach().clone() for p in r.bucket.params], None if r._cap is None else len(r._cap.graphs))
        (l0, w0, _), (l1, w1, n_seg) = out[False], out[True]
        assert n_seg == 3, n_seg                             # forward up to the normalisers | rest of the step up to the bucket | Adam
        assert l0 == l1, (l0, l1)
        assert all(torch.equal(a, b) for a, b in zip(w0, w1))
        if rank == 0:
            q.put(dict(ok=True, losses=l1, segments=n_seg))
    finally:
        dist.destroy_process_group()


def test_geo_runner_graph_replays_the_data_parallel_step(tmp_path):
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_graph_worker, args=(rk, 2, port, str(tmp_path), q)) for rk in range(2)]
    for p in procs:
        p.start()
    try:
        t0 = time.time()
        while q.empty():
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), 'a rank died'
            assert time.time() - t0 < 240, 'ranks did not finish'
            time.sleep(0.2)
        got = q.get()
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert got['ok'] and got['segments'] == 3 and len(set(got['losses'])) > 1


def _decomp_model():
    from oracle import decomp as od
    from tests.decomp_util import make_config, load_oracle_params
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, _ = od.make_model_params(seed=0, K=15)
    model = load_oracle_params(get_model_class('vq_nfr')(make_config(n_rays_per_step=128)), p, 'cuda')
    model.get_codebook(); _ = model.light
    return model


def _decomp_worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import decomp as od
    from tests.decomp_util import make_batch
    from vqnerf_release_amd import parallel
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        model = _decomp_model()
        opt = torch.optim.Adam(model.trainable_variables, lr=5e-4, eps=1e-7, amsgrad=True)
        tr = train_nfr.Trainer(model, opt)
        full = make_batch(od.make_points(256, seed=9), 'cuda')
        lo, hi = parallel.shard_range(256)
        half = tuple(t[lo:hi].contiguous() if torch.is_tensor(t) else t for t in full)
        wl, _, _ = tr.train_iter(half, global_bs=256)
        parallel.assert_replicas_identical(list(model.trainable_variables), 'reflectance model after the step')
        if rank == 0:
            q.put(dict(loss=float(wl), codebook=model._codebook.detach().cpu().numpy().copy(),
                       counts=model.vq_layer.ema_cluster_size.hidden.detach().cpu().numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_decomp_trainer_dp2_shares_codebook_statistics():
    """The VQ stage under data parallelism: EMA statistics (code counts, x^T . onehot) are summed over the ranks before the
    codebook moves, so both replicas hold the same codebook -- and it is the one a single process computes from all rows."""
    import time
    import torch.multiprocessing as mp
    from oracle import decomp as od
    from tests.decomp_util import make_batch
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_decomp_worker, args=(rk, 2, port, q)) for rk in range(2)]
    for p in procs:
        p.start()
    try:
        t0 = time.time()
        while q.empty():
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), 'a rank died'
            assert time.time() - t0 < 240, 'ranks did not finish'
            time.sleep(0.2)
        got = q.get()
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    model = _decomp_model()
    opt = torch.optim.Adam(model.trainable_variables, lr=5e-4, eps=1e-7, amsgrad=True)
    tr = train_nfr.Trainer(model, opt)
    wl, _, _ = tr.train_iter(make_batch(od.make_points(256, seed=9), 'cuda'), global_bs=256)
    np.testing.assert_array_equal(got['counts'], model.vq_layer.ema_cluster_size.hidden.cpu().numpy())        # integer counts: exact
    np.testing.assert_allclose(got['codebook'], model._codebook.detach().cpu().numpy(), rtol=2e-5, atol=2e-6)
    assert abs(got['loss'] - float(wl)) <= 5e-3 * max(1.0, abs(float(wl)))     # batch-level terms (pair similarity, code spread) are per rank


def _decomp_graph_worker(rank, world, port, q):
    import torch.distributed as dist
    from oracle import decomp as od
    from tests.decomp_util import make_batch, make_config
    from vqnerf_release_amd import parallel
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        cfg = make_config(n_rays_per_step=128)
        batches = []
        for i in range(6):
            full = make_batch(od.make_points(256, seed=40 + i), 'cuda')
            lo, hi = parallel.shard_range(256)
            batches.append(tuple(t[lo:hi].contiguous() if torch.is_tensor(t) else t for t in full))
        runs = {}
        for graph in (False, True):
            model = _decomp_model()
            opt, _, clip = train_nfr.make_optimizer(cfg, model.trainable_variables, capturable=True)
            tr = train_nfr.Trainer(model, opt, clip=clip, graph=graph)
            losses = [float(tr.train_iter(b, global_bs=256)[0]) for b in batches]
            parallel.assert_replicas_identical(list(model.trainable_variables) + [model._codebook], 'replicas (graph=%s)' % graph)
            runs[graph] = (losses, [v.detach().clone() for v in model.trainable_variables], model._codebook.detach().clone(),
                           None if tr._captured is None else (len(tr._captured.graphs), [w for _, _, w in tr._captured.exchanges]))
        (l0, w0, c0, _), (l1, w1, c1, segs) = runs[False], runs[True]
        same = l0 == l1 and all(torch.equal(a, b) for a, b in zip(w0, w1)) and torch.equal(c0, c1)
        if rank == 0:
            q.put(dict(same=bool(same), segs=segs, losses=(l0, l1)))
    finally:
        dist.destroy_process_group()


def test_decomp_trainer_graph_replays_the_data_parallel_step():
    """Trainer(graph=True) under data parallelism: the captured step is three HIP graphs with the two collectives (VQ statistics,
    gradient bucket) issued eagerly between them -- parameters, codebook and losses after six steps are bit for bit those of the
    eager data-parallel trainer, on both ranks."""
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_decomp_graph_worker, args=(rk, 2, port, q)) for rk in range(2)]
    for p in procs:
        p.start()
    try:
        t0 = time.time()
        while q.empty():
            assert all(p.is_alive() or p.exitcode == 0 for p in procs), 'a rank died'
            assert time.time() - t0 < 300, 'ranks did not finish'
            time.sleep(0.2)
        got = q.get()
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
    assert got['segs'] == (3, ['all_reduce:vq_stats', 'all_reduce:grad_bucket']), got['segs']
    assert got['same'], got['losses']
