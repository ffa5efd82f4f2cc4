"""GPU: RCCL itself under the trainers' collectives.  A gpurun box has ONE card and RCCL refuses two ranks on one device ("Duplicate GPU
detected", profiles/r05_engine_diag.txt #3), so every multi-rank test of this repo runs on gloo.  What CAN run here is a one-rank RCCL
communicator: `torch.distributed` backend "nccl" (= RCCL on ROCm), world size 1 -- the same library calls, streams and buffers as on a
node, minus the wire.  In a child process (a process group is process-global state):
  * the flat gradient bucket of the reflectance trainer and the VQ statistics buffer go through `parallel.all_reduce_sum` (values
    unchanged by a one-rank sum; RCCL's own kernel runs on the buffer, on the trainer's stream);
  * a step recorded as HIP-graph SEGMENTS around its collectives (parallel.SegmentedCapture) replays with the REAL RCCL all-reduce between
    the graphs and ends bit-identical to the eager data-parallel step -- the validation `bench.py` runs before it times the graph legs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import bench
from vqnerf_release_amd import parallel
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
out = {'backend': dist.get_backend(), 'world': dist.get_world_size()}
# 1. the trainers' two buffers through RCCL
model, tr, step = bench.decomp_train_setup(dev, 0, 1)
for _ in range(3):
    step()
flat = tr.bucket.flat
before = flat.clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    parallel.all_reduce_sum(flat, what='all_reduce:grad_bucket')
torch.cuda.synchronize()
out['bucket_bytes'] = int(flat.numel() * 4)
out['bucket_all_reduce_us'] = (time.perf_counter() - t0) / 20 * 1e6
out['bucket_unchanged'] = bool(torch.equal(flat, before))
stats = torch.arange(257 * 15, dtype=torch.float32, device=dev)
s0 = stats.clone()
parallel.all_reduce_sum(stats, what='all_reduce:vq_stats')
out['stats_unchanged'] = bool(torch.equal(stats, s0))
# 2. graph segments with the real collective between them: SegmentedCapture cuts at all_reduce_sum
x = torch.randn(1 << 16, device=dev)
buf = torch.zeros_like(x)
y = torch.zeros_like(x)
def body():
    buf.copy_(x * 2.0)
    parallel.all_reduce_sum(buf, what='all_reduce:test')          # (recording: ends graph 0, starts graph 1)
    y.copy_(buf + 1.0)
body()
torch.cuda.synchronize()
want = y.clone()
y.zero_(); buf.zero_()
cap = parallel.SegmentedCapture()
with cap:
    body()
cap.replay()
torch.cuda.synchronize()
out['graph_segments'] = len(cap.graphs)
out['segment_replay_equal'] = bool(torch.equal(y, want))
# 3. bench.py's own validation of the graph-segment DP step against the eager one, on this backend
#    (world 1: nothing cuts, but every collective call site runs under RCCL's process group)
rep = bench.dp_graph_selfcheck(dev, 0, 1, 'nccl', steps=2)
out['dp_graph_selfcheck'] = {k: rep.get(k) for k in ('bit_identical', 'captured', 'graph_segments', 'error')}
dist.destroy_process_group()
print('RESULT ' + json.dumps(out))
'''


def test_one_rank_rccl_group_carries_the_trainers_collectives(tmp_path):
    import socket
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-c', CHILD % {'root': ROOT}], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1]
    out = json.loads(line[7:])
    print(out)
    assert out['backend'] == 'nccl' and out['world'] == 1
    assert out['bucket_unchanged'] and out['stats_unchanged'] and out['bucket_bytes'] > 3_000_000
    assert out['graph_segments'] == 2 and out['segment_replay_equal']
    assert out['dp_graph_selfcheck']['error'] is None and out['dp_graph_selfcheck']['bit_identical'] is True
    try:
        os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(ROOT, 'gpurun_out', 'rccl_single_rank.json'), 'w') as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
