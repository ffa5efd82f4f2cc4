"""Full-size geometry networks (the two-image training kernels): 120 eager + 120 replayed steps, losses finite, memory flat."""
import sys, time
sys.path.insert(0, '.')
import torch, bench
dev = torch.device('cuda:0')
for graph in (False, True):
    runner, step = bench.geo_train_setup(dev, 0, 2560, graph=graph)
    hist, mem = [], []
    t0 = time.time()
    for it in range(120):
        st = step()
        if it % 20 == 19:
            torch.cuda.synchronize()
            hist.append(round(float(st['loss']), 4))
            mem.append(torch.cuda.memory_reserved() / 2**30)
    torch.cuda.synchronize()
    ok = all(torch.isfinite(p).all() for p in list(runner.sdf_network.parameters()) + list(runner.color_network.parameters()))
    print(f'graph={graph}: 120 steps in {time.time()-t0:.1f} s; reserved GiB every 20 steps: ' + ' '.join(f'{m:.1f}' for m in mem), '| params finite:', bool(ok), '| losses', hist)
    del runner, step
    torch.cuda.empty_cache()
