"""CPU, world_size 2 over gloo: the data-parallel plumbing (vqnerf_release_amd/parallel.py).
  * one flat bucket [grads || extras] -> one all-reduce; replicas stay bit-identical;
  * DP over two half-batches == single process over the full batch (loss normalised by the GLOBAL batch);
  * VQ EMA statistics reduced across ranks give the same codebook update as one process would."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vqnerf_release_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net(seed=0):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def _data(n=64):
    g = np.random.default_rng(1)
    return torch.tensor(g.normal(size=(n, 6)), dtype=torch.float32), torch.tensor(g.normal(size=(n, 3)), dtype=torch.float32)


def _ema_update(counts, dw, C, eps=1e-5, decay=0.999):
    # first EMA step: zero-debiased average == the value itself (hidden = v (1-decay); / (1 - decay^1))
    cs, K = counts.clone(), counts.numel()
    n = cs.sum()
    cs = (cs + eps) / (n + K * eps) * n
    used = (counts > 0).float()
    return dw / cs[None, :] * used[None, :] + C * (1 - used[None, :])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        assert parallel.is_dist() and parallel.world_size() == world and parallel.rank() == rank
        x, y = _data()
        lo, hi = parallel.shard_range(x.shape[0])
        net = _net()
        opt = torch.optim.Adam(net.parameters(), lr=1e-2)
        bucket = parallel.FlatBucket(net.parameters(), n_extra=2)
        for step in range(3):
            opt.zero_grad(set_to_none=True)
            bucket.attach()
            per_example = ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean(-1)
            loss = per_example.sum() / x.shape[0]                    # global batch, as compute_average_loss
            loss.backward()
            with torch.no_grad():
                bucket.extra[0] = loss
                bucket.extra[1] = float(hi - lo)
            extra = bucket.all_reduce()
            assert float(extra[1]) == x.shape[0]
            opt.step()
            parallel.assert_replicas_identical(list(net.parameters()))
        # the same three steps with the gradient all-reduce cut into slices that leave as soon as their last gradient is in (overlap
        # with the backward tail): bit-identical parameters, extras intact, unused parameters handled
        net2 = _net()
        dead = torch.nn.Parameter(torch.ones(3))                       # never used: its slice must still go out (as zeros)
        opt2 = torch.optim.Adam(list(net2.parameters()) + [dead], lr=1e-2)
        bucket2 = parallel.FlatBucket(list(net2.parameters()) + [dead], n_extra=2).enable_overlap(n_buckets=3)
        assert len(bucket2._slices) >= 2 and bucket2._slices[-1]['hi'] == bucket2.n_grad
        for step in range(3):
            opt2.zero_grad(set_to_none=True)
            if step == 1:
                bucket2.attach()                                       # both styles: grads as views of the bucket, or fresh tensors
            loss2 = ((net2(x[lo:hi]) - y[lo:hi]) ** 2).mean(-1).sum() / x.shape[0]
            loss2.backward()
            assert len(bucket2._pending) >= 1                          # slices really left during backward
            with torch.no_grad():
                bucket2.extra[0] = loss2
                bucket2.extra[1] = float(hi - lo)
            extra2 = bucket2.all_reduce()
            assert float(extra2[1]) == x.shape[0] and not bucket2._pending
            assert float(bucket2.views[-1].abs().sum()) == 0.0             # (no gradient: zeros went out, Adam skips it)
            opt2.step()
        for a, b in zip(net.parameters(), net2.parameters()):
            assert torch.equal(a, b)
        assert float(extra2[0]) == float(extra[0])
        # VQ statistics: local one-hot stats of this rank's rows, reduced
        g = np.random.default_rng(2)
        z = torch.tensor(g.uniform(0, 1, (40, 8)), dtype=torch.float32)
        C = torch.tensor(g.uniform(0, 1, (8, 5)), dtype=torch.float32)
        zl = z[slice(*parallel.shard_range(40))]
        idx = torch.cdist(zl, C.t()).argmin(1)
        enc = torch.nn.functional.one_hot(idx, 5).float()
        counts, dw = parallel.VQStatsReducer()(enc.sum(0), zl.t() @ enc)
        upd = _ema_update(counts, dw, C)
        parallel.assert_replicas_identical([upd], 'codebook update')
        if rank == 0:
            q.put(dict(params=[p.detach().numpy().copy() for p in net.parameters()], loss=float(extra[0]), upd=upd.numpy().copy()))
    finally:
        dist.destroy_process_group()


def test_dp2_equals_single_process():
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    # single-process reference over the full batch
    x, y = _data()
    net = _net()
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    for step in range(3):
        opt.zero_grad()
        loss = ((net(x) - y) ** 2).mean(-1).sum() / x.shape[0]
        loss.backward()
        opt.step()
    for a, b in zip(got['params'], net.parameters()):
        torch.testing.assert_close(torch.tensor(a), b.detach(), rtol=1e-5, atol=1e-6)
    assert abs(got['loss'] - float(loss)) < 1e-5
    g = np.random.default_rng(2)
    z = torch.tensor(g.uniform(0, 1, (40, 8)), dtype=torch.float32)
    C = torch.tensor(g.uniform(0, 1, (8, 5)), dtype=torch.float32)
    enc = torch.nn.functional.one_hot(torch.cdist(z, C.t()).argmin(1), 5).float()
    torch.testing.assert_close(torch.tensor(got['upd']), _ema_update(enc.sum(0), z.t() @ enc, C), rtol=1e-6, atol=1e-6)


def test_single_process_paths_are_noops():
    assert not parallel.is_dist() and parallel.world_size() == 1 and parallel.rank() == 0
    assert parallel.shard_range(10, 1, 3) == (4, 7) and parallel.shard_range(10, 0, 3) == (0, 4)
    net = _net()
    b = parallel.FlatBucket(net.parameters(), n_extra=1).attach()
    net(torch.ones(2, 6)).sum().backward()
    flat_before = b.flat.clone()
    b.all_reduce()
    assert torch.equal(flat_before, b.flat)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(b.params, b.views))     # backward wrote into the bucket
    c, d = parallel.VQStatsReducer()(torch.ones(3), torch.ones(2, 3))
    assert c.sum() == 3 and d.sum() == 6
