"""Geo training step (2560 rays): wall ms per step (eager / graph) and the per-kernel clock of one eager step."""
import sys, time
sys.path.insert(0, '.')
import torch, bench
from vqnerf_release_amd import _C
dev = torch.device('cuda:0')
for graph in (False, True):
    runner, step = bench.geo_train_setup(dev, 0, 2560, graph=graph)
    for _ in range(6): step()
    torch.cuda.synchronize()
    ws = []
    for w in range(3):
        t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); ws.append((time.perf_counter() - t0) / 10 * 1e3)
    print(f'graph={graph}: ms/step three windows', ['%.2f' % x for x in ws], flush=True)
    if not graph:
        _C.KernelClock.reset(True)
        step(); torch.cuda.synchronize()
        summ = _C.KernelClock.summary(); _C.KernelClock.reset(False)
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])[:10]:
            print('      ', k, v)
    del runner, step
