"""Where does a process's one-off ~100 ms stall after ~4,100 kernel launches come from (bench.py primes training legs past it)?
Batches of 100 launches, each batch timed on the host with a sync; three kinds of launch in separate processes:
  torch   : x.add_(1) on a 1-element tensor            (framework launch path)
  ctypes  : vqn_clip_preserve on a 1-element tensor    (this library's launch path, no torch kernel)
  events  : torch launches with a pair of cuda events recorded around each (what _C.KernelClock does when enabled)"""
import subprocess, sys, time
if len(sys.argv) == 1:
    for kind in ('torch', 'ctypes', 'events'):
        print(subprocess.run([sys.executable, __file__, kind], capture_output=True, text=True).stdout, flush=True)
    sys.exit(0)
sys.path.insert(0, '.')
import torch
from vqnerf_release_amd import _C
kind = sys.argv[1]
x = torch.zeros(1, device='cuda')
torch.cuda.synchronize()
slow, total, t_all = [], 0, time.perf_counter()
evs = []
for b in range(120):
    t0 = time.perf_counter()
    for _ in range(100):
        if kind == 'torch':
            x.add_(1.0)
        elif kind == 'ctypes':
            _C.clip_preserve(x, 0.0, 1.0)
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); x.add_(1.0); e1.record()
            evs.append((e0, e1))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    total += 100
    if dt > 8.0:
        slow.append((total, round(dt, 1)))
print(f'{kind}: 12,000 launches in {time.perf_counter() - t_all:.2f} s; batches of 100 slower than 8 ms (launch count at batch end, ms):', slow)
