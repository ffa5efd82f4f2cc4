"""CPU, world_size 2 over gloo: the reference's ONLY data-parallel site -- the stage-1 / stage-3 epoch loop of
decomp/nerfvq_nfr3/nerfactor/trainvali.py:436-486 (MirroredStrategy: per-replica step, loss normalised by the GLOBAL batch, summed
gradients) -- as `train_nfr.fit_stage` runs it here: two ranks, each on its half of every step's pair sample, end with the parameters
(and per-epoch losses) of ONE process that took the whole samples, for `nfr_unit` (stage 1, with a pretrain epoch) and `ref_nfr`
(stage 3: frozen encoder / specular head, constant light).  VERDICT r03 missing #3.  Torch statements of the model on CPU tensors (the
HIP side of the same step is tests/test_gpu_parallel.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_PER_RANK, EPOCHS, VIEWS = 24, 3, 2


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cfg(model, n):
    from tests.decomp_util import make_config
    return make_config(model=model, mlp_width=32, conv_width=32, n_freqs_xyz=4, light_h=4, n_rays_per_step=n // 2, epochs=EPOCHS,
                       pretrain_epochs=1, ckpt_period=1000, vali_period=1000, lr=2e-3, random_seed=3)


def _view(model_name, v, n):
    """A fixed, all-foreground pair sample of n rows for view v (what outer_sample would hand to the step)."""
    rng = np.random.default_rng(100 + v)
    xyz = rng.uniform(-1, 1, (n, 3)); xyz /= np.linalg.norm(xyz, axis=1, keepdims=True)
    normal = xyz + 0.1 * rng.normal(size=(n, 3)); normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
    alpha = torch.ones(n, 1)
    batch = (['v%d' % v] * n, torch.zeros(n, 2), T(np.tile([[0.0, 0.0, 4.0]], (n, 1))), torch.zeros(n, 3), T(rng.uniform(0, 1, (n, 3))), alpha,
             alpha.clone(), T(0.8 * xyz), T(normal))
    if model_name == 'ref_nfr':
        batch = batch + (T(rng.uniform(0, 1, (n, 3))),)                                    # the per-point reference colour (ref_nfr.py:180-184)
    return batch + (T((rng.uniform(size=(n, 32)) < 0.7).astype(np.float32)),)             # lvis [n, light_h * 2 light_h]


class _Views:
    """duck-typed stand-in for datasets/shape_unit.Dataset: `bs` pair rows per step and rank, a fixed list of views"""

    def __init__(self, bs):
        self.bs = bs

    def build_pipeline(self, no_shuffle=False):
        return list(range(VIEWS))

    def get_n_views(self):
        return VIEWS


def _run(model_name, rank, world, outdir):
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    n_all = N_PER_RANK * 2
    cfg = _cfg(model_name, n_all if world == 1 else N_PER_RANK)
    torch.manual_seed(0)
    model = get_model_class(model_name)(cfg)
    model.build_nets(device='cpu', seed=5)
    if model_name == 'ref_nfr':                                                            # stage hand-off: frozen stage-2 parts
        for name in ('fine_enc', 'bottleneck', 'spec_out'):
            for p in model.net[name].parameters():
                p.requires_grad_(False)
        model.register_trainable()

    def sample(view, config, data_type, generator=None, neighbour=None):
        full = _view(model_name, view, n_all)
        if world == 1:
            return full
        lo, hi = rank * N_PER_RANK, (rank + 1) * N_PER_RANK                               # pairs stay together: N_PER_RANK is even
        return tuple(t[lo:hi] for t in full)
    real = train_nfr.outer_sample
    train_nfr.outer_sample = sample
    try:                                         # (restored: the single-process leg runs inside the test process, other tests follow it)
        model, hist = train_nfr.fit_stage(cfg, outdir, _Views(n_all if world == 1 else N_PER_RANK),
                                          None, model=model, device='cpu', log=lambda *_: None)
    finally:
        train_nfr.outer_sample = real
    return model, hist


def _worker(rank, world, port, model_name, outdir, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from vqnerf_release_amd import parallel
        model, hist = _run(model_name, rank, world, os.path.join(outdir, 'dp'))
        parallel.assert_replicas_identical([p for p in model.parameters()])
        if rank == 0:
            q.put(dict(params={k: v.detach().numpy().copy() for k, v in model.state_dict().items()}, loss=hist['loss']))
    except Exception:
        import traceback
        q.put(dict(error='rank %d: %s' % (rank, traceback.format_exc())))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('model_name', ['nfr_unit', 'ref_nfr'])
def test_fit_stage_on_two_ranks_is_the_single_process_run(model_name, tmp_path):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, model_name, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get()
    assert 'error' not in got, got.get('error')
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    model, hist = _run(model_name, 0, 1, str(tmp_path / 'single'))
    want = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    assert got['params'].keys() == want.keys()
    moved = 0
    init = {k: v.detach().numpy() for k, v in _fresh(model_name).state_dict().items()}
    for k in want:
        np.testing.assert_allclose(got['params'][k], want[k], rtol=2e-4, atol=2e-6, err_msg=k)
        moved += int(np.abs(want[k] - init[k]).max() > 1e-5)
    assert moved >= 4, 'the run did not train'
    # the step's loss travels in the bucket's extras: every rank reports the GLOBAL loss, i.e. the single process's
    assert len(got['loss']) == EPOCHS and all(np.isfinite(got['loss']))
    np.testing.assert_allclose(got['loss'], hist['loss'], rtol=1e-5)
    assert hist['loss'][-1] < hist['loss'][0]
    if model_name == 'ref_nfr':                                                            # frozen parts did not move
        for k in want:
            if k.split('.')[1] in ('fine_enc', 'bottleneck', 'spec_out') if k.startswith('net.') else False:
                np.testing.assert_array_equal(want[k], init[k])


def _fresh(model_name):
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    m = get_model_class(model_name)(_cfg(model_name, 2 * N_PER_RANK))
    m.build_nets(device='cpu', seed=5)
    _ = m.light
    return m
