"""Framework (non-HIP-library) kernels of the 262,144-point reflectance step by aten op and input shapes (torch profiler)."""
import sys, collections
sys.path.insert(0, '.')
import numpy as np, torch, bench
from torch.profiler import profile, ProfilerActivity
exec(open('scripts/debug/decomp256k_time.py').read().split("for _ in range(2): tr.train_iter")[0])
for _ in range(4): tr.train_iter(batch, global_bs=n)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.train_iter(batch, global_bs=n)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.device_time_total > 0 and e.key.startswith('aten::'):
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('aten self device time total (us):', tot)
for t, c, k, sh in rows[:45]:
    print(f'{t:9.0f} us  x{c:<3d} {k:28s} {sh}')
