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
