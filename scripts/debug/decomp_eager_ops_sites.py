"""Which Python lines issue the torch ops of one EAGER reflectance training step (2048 points), backward thread included?
torch profiler with stacks: every aten op that is a direct child of Python code, grouped by the innermost frame inside the package."""
import sys, collections
sys.path.insert(0, '.')
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
model, tr, step = bench.decomp_train_setup(dev, 0, 1, graph=False)
for _ in range(4): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
    torch.cuda.synchronize()
sites = collections.Counter()
skip = ('aten::view', 'aten::reshape', 'aten::detach', 'aten::alias', 'aten::expand', 'aten::slice', 'aten::select', 'aten::unsqueeze', 'aten::squeeze',
        'aten::as_strided', 'aten::t', 'aten::transpose', 'aten::permute', 'aten::empty', 'aten::size', 'aten::stride', 'aten::item', 'aten::_local_scalar',
        'aten::is_', 'aten::resize', 'aten::lift', 'aten::result_type', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::unbind', 'aten::narrow', 'aten::flatten',
        'aten::empty_like', 'aten::empty_strided', 'aten::set_', 'aten::unflatten', 'aten::chunk', 'aten::split', 'aten::_unsafe_view', 'aten::numel', 'aten::dim')
for e in prof.events():
    if not e.name.startswith('aten::') or e.name in skip or any(e.name.startswith(s) for s in skip):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith('aten::'):
        continue                                   # only top-level ops
    site = 'autograd-engine / no package frame'
    for fr in (e.stack or []):
        if 'vqnerf_release_amd/' in fr:
            site = fr.split('vqnerf_release_amd/')[-1]
            break
    sites[(site, e.name)] += 1
by = collections.Counter()
for (s, n), c in sites.items(): by[s] += c
print('top-level aten ops in one eager step:', sum(sites.values()))
for s, c in by.most_common(80):
    ops = ', '.join(f'{n.replace("aten::", "")}x{k}' for (ss, n), k in sorted(sites.items(), key=lambda kv: -kv[1]) if ss == s)[:150]
    print(f'{c:4d}  {s[:70]:70s} {ops}')
