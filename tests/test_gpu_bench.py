"""GPU: bench.py's multi-rank paths rehearsed with 2 ranks on ONE card (gloo stands in for RCCL: `VQN_BENCH_BACKEND=gloo`;
the driver launches the real thing with one rank per GPU).  Checks the contract of the JSON line and that the data-parallel
training legs really run their collectives (the gradient bucket + the VQ statistics all-reduce) on both ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(extra_args, nproc=2, want_rc=0, launcher=True, detail=False, **env_extra):
    """launcher=True: the driver's own command (python -m torch.distributed.run ... bench.py --gpus N); False: plain `python bench.py
    --gpus N` with no WORLD_SIZE -- bench.py starts its ranks itself.  Returns the parsed stdout line (and the sidecar if asked)."""
    import tempfile
    side = os.path.join(tempfile.mkdtemp(prefix='vqn_bench_test_'), 'bench_detail.json')
    env = dict(os.environ, VQN_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', VQN_BENCH_DETAIL=side, **env_extra)
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    head = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={nproc}', '--master-addr', '127.0.0.1',
            '--master-port', str(_free_port())] if launcher else [sys.executable]
    cmd = head + [os.path.join(ROOT, 'bench.py'), '--gpus', str(nproc)] + extra_args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == want_rc, (r.returncode, r.stdout[-3000:] + r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]                    # ONE JSON line, from rank 0
    assert len(lines[0].encode()) < 8192                        # ... that the driver can capture
    line = json.loads(lines[0])
    if detail:
        with open(side) as f:
            return line, json.load(f)
    return line


def test_two_rank_render_line_and_dp_training_legs():
    res = _run(['--steps', '2', '--warmup', '1', '--rays', '16000'])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
        assert k in res, k
    assert res['n_gpus'] == 2 and res['scaling'] == 'weak' and res['unit'] == 'rays/s' and res['vs_baseline'] is None
    assert abs(res['value'] - 2 * 16000 * 2 / (res['ms_per_step'] * 2 * 1e-3)) < 1e-6 * res['value']     # whole job / max-over-ranks time
    assert 0.05 < res['roofline']['frac'] < 1.0 and res['roofline']['bound'] == 'mfma'
    assert 'cpu_baseline' not in res                             # rank 0 at N = 1 only
    dp = res['extra']['dp_train']
    assert dp['n_ranks_seen'] == 2
    g, d = dp['geo'], dp['decomp']
    assert g['last_train_backend'] == 'hip'
    assert g['all_reduce']['grad_bucket']['calls_per_step'] == 1 and g['all_reduce']['loss_normalisers']['calls_per_step'] == 1
    assert d['all_reduce']['grad_bucket']['calls_per_step'] == 1 and d['all_reduce']['vq_stats']['calls_per_step'] == 1
    assert g['all_reduce']['total_us_per_step'] > 0 and d['all_reduce']['total_us_per_step'] > 0
    assert g['grad_bucket_bytes'] > 3_000_000 and d['grad_bucket_bytes'] > 3_000_000           # 0.80 M and 0.78 M fp32 gradients (+ extras)
    assert d['replicas_bit_identical_after_steps'] is True
    assert g['rays_per_s'] > 0 and d['points_per_s'] > 0
    dg = dp['decomp_graph']
    assert 'error' not in dg and dg['graph_segments'] == 3 and dg['all_reduce']['vq_stats']['calls_per_step'] == 1


def test_two_rank_headline_survives_stuck_dp_legs():
    """The data-parallel legs have never run on a multi-GPU RCCL node: if they do not return (here: a 1-second limit), rank 0 still prints
    the complete render line, with the error under extra.dp_train, and every rank leaves with a NON-ZERO exit code (3): a hang of
    GPU-touching ranks must not read as success to the launcher."""
    res = _run(['--steps', '2', '--warmup', '1', '--rays', '16000'], want_rc=1, VQN_BENCH_DP_TIMEOUT='1')     # (torchrun reports its ranks' failure as 1)
    assert res['n_gpus'] == 2 and res['value'] > 0 and 0.05 < res['roofline']['frac'] < 1.0
    assert 'did not return within 1 s' in res['extra']['dp_train']['error']
    assert res['exit_code'] == 3 and res['extra']['dp_train']['exit_code'] == 3


def test_plain_python_bench_gpus_2_launches_its_own_ranks():
    """VERDICT r04 missing #2: `python3 bench.py --gpus 2` with WORLD_SIZE unset (how `--gpus 1` is started) must run -- bench.py starts
    `python -m torch.distributed.run` as a child before touching the GPU, relays rank 0's line and leaves with the child's code."""
    res, detail = _run(['--steps', '2', '--warmup', '1', '--rays', '16000'], launcher=False, detail=True)
    assert res['n_gpus'] == 2 and res['config']['parallelism'].startswith('views x2')
    dp = res['extra']['dp_train']
    assert dp['n_ranks_seen'] == 2 and dp['backend'] == 'gloo'
    assert dp['geo']['all_reduce']['grad_bucket']['us_per_step'] > 0 and dp['decomp']['all_reduce']['vq_stats']['us_per_step'] > 0
    assert dp['dp_graph_selfcheck']['bit_identical'] is True
    assert 'traffic_note' in detail['roofline'] and detail['extra']['dp_train']['n_ranks_seen'] == 2      # the prose went to the sidecar


def test_two_rank_train_mode_times_the_dp_step():
    res = _run(['--mode', 'train', '--steps', '3', '--warmup', '1'])
    assert res['n_gpus'] == 2 and res['config']['n_ranks_seen'] == 2 and res['config']['parallelism'] == 'dp2'
    assert abs(res['value'] - 2 * 2560 / (res['ms_per_step'] * 1e-3)) < 1e-6 * res['value']
    assert res['all_reduce']['grad_bucket']['calls_per_step'] == 1 and res['all_reduce']['total_us_per_step'] > 0
    assert res['last_train_backend'] == 'hip' and res['roofline']['bound'] == 'mfma'
    assert any(k.startswith(('vqn_tile_program', 'vqn_neus_train_', 'vqn_refl_train_')) for k in res['kernel_ms_per_step'])
    assert res['extra']['decomp_train_dp']['all_reduce']['vq_stats']['calls_per_step'] == 1
