"""Which ops issue device-to-device memcpys in the CAPTURED reflectance step?  torch profiler around the capturing call:
runtime hipMemcpyAsync events, attributed to the innermost CPU op / python frame that contains them in time."""
import sys, collections
sys.path.insert(0, '.')
import torch, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
model, tr, step = bench.decomp_train_setup(dev, 0, 1, graph=True)
for it in range(5):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    ev = list(prof.events())
    rt = [e for e in ev if e.name.startswith('hipMemcpy')]
    print('step', it, 'captured', tr._captured is not None, 'hipMemcpy* runtime calls:', len(rt))
    if len(rt) > 20:
        cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU and not e.name.startswith('hip')]
        by = collections.Counter()
        for m in rt:
            t = m.time_range.start
            best = None
            for e in cpu:
                if e.time_range.start <= t <= e.time_range.end and (best is None or e.time_range.elapsed_us() < best.time_range.elapsed_us()):
                    best = e
            frames = [s for s in (best.stack or []) if 'vqnerf_release_amd' in s][:2] if best is not None else []
            by[(best.name if best is not None else '?', ' <- '.join(frames))] += 1
        for (n, s), c in by.most_common(12):
            print('   ', c, n, '|', s)
        chains = collections.Counter()
        for m in rt:
            t = m.time_range.start
            cont = sorted([e for e in cpu if e.time_range.start <= t <= e.time_range.end], key=lambda e: -e.time_range.elapsed_us())
            chains[' > '.join(e.name[:60] for e in cont[-4:])] += 1
        for ch, c in chains.most_common(15):
            print('   ', c, ch)
