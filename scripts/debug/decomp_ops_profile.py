"""Which Python lines launch the small kernels of one reflectance training step (2048 points)?  torch.profiler with stacks."""
import sys, collections
sys.path.insert(0, '.')
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
model, tr, step = bench.decomp_train_setup(dev, 0, 1, graph=False)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
by_site = collections.Counter()
by_op = collections.Counter()
for ev in prof.events():
    if ev.device_type.name == 'CPU' and ev.name.startswith('aten::') and ev.cpu_parent is not None and not ev.cpu_parent.name.startswith('aten::'):
        pass
n_kernels = 0
for ev in prof.events():
    if str(ev.device_type).endswith('CUDA'):
        n_kernels += 1
# attribute device kernels to the innermost repo frame of the launching CPU op
for ev in prof.events():
    if ev.device_type.name != 'CPU' or not ev.kernels:
        continue
    site = 'unknown'
    for fr in (ev.stack or []):
        if '/root/repo/' in fr or 'vqnerf_release_amd' in fr or 'bench.py' in fr:
            site = fr.split('/root/repo/')[-1] if '/root/repo/' in fr else fr
            break
    by_site[site] += len(ev.kernels)
    by_op[ev.name] += len(ev.kernels)
print('device kernels in one step:', sum(by_site.values()))
for s, c in by_site.most_common(45):
    print(f'{c:5d}  {s[:150]}')
print('--- by op')
for s, c in by_op.most_common(25):
    print(f'{c:5d}  {s}')
