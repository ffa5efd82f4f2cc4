"""Captured 2048-point reflectance step: ms per replay (three windows of 40 replays, all listed)."""
import sys, time
sys.path.insert(0, '.')
import torch, bench
dev = torch.device('cuda:0')
model, tr, step = bench.decomp_train_setup(dev, 0, 1, graph=True)
for _ in range(8): step()
torch.cuda.synchronize()
ws = []
for w in range(3):
    t0 = time.perf_counter()
    for _ in range(40): step()
    torch.cuda.synchronize()
    ws.append((time.perf_counter() - t0) / 40 * 1e3)
print('captured', tr._captured is not None, 'ms per replay, three windows:', ['%.4f' % x for x in ws])
