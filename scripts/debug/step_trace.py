"""Per-step wall time of the first 40 geo training steps (eager, bench's own loop): where does the one-off stall land, and what is it?"""
import sys, time, os
sys.path.insert(0, '.')
import torch, bench
dev = torch.device('cuda:0')
runner, step = bench.geo_train_setup(dev, 0, 2560, graph=False)
ts, mem = [], []
for it in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    step()
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    mem.append(torch.cuda.memory_reserved() / 2**30)
print('ms per step :', ' '.join(f'{t:.0f}' for t in ts))
print('reserved GiB:', ' '.join(f'{m:.0f}' for m in mem))
st = torch.cuda.memory_stats()
print('num_alloc_retries', st.get('num_alloc_retries'), 'segments', st.get('segment.all.current'), 'hipMalloc calls (num_device_alloc)', st.get('num_device_alloc'), 'frees', st.get('num_device_free'))
