"""Full-size geometry networks (the exact-split training kernels): 400 eager + 400 replayed steps with a real learning rate; the
loss falls, nothing goes non-finite, memory stays flat, eager and replayed runs see the same losses."""
import sys, time
sys.path.insert(0, '.')
import torch, bench
dev = torch.device('cuda:0')
out = {}
for graph in (False, True):
    runner, step = bench.geo_train_setup(dev, 0, 2560, graph=graph)
    runner.warm_up_end = 10                                   # (reach the full learning rate quickly)
    hist, mem = [], []
    t0 = time.time()
    for it in range(400):
        st = step()
        if it % 50 == 49:
            torch.cuda.synchronize()
            hist.append(round(float(st['loss']), 5))
            mem.append(torch.cuda.memory_reserved() / 2**30)
    torch.cuda.synchronize()
    ok = all(bool(torch.isfinite(p).all()) for p in list(runner.sdf_network.parameters()) + list(runner.color_network.parameters()))
    print(f'graph={graph}: 400 steps in {time.time()-t0:.1f} s; reserved GiB: ' + ' '.join(f'{m:.0f}' for m in mem), '| params finite:', ok, '| loss every 50:', hist)
    out[graph] = hist
    assert ok and all(h == h for h in hist) and hist[-1] < hist[0]
    del runner, step
    torch.cuda.empty_cache()
print('eager vs replayed losses equal:', out[False] == out[True])
