import sys, time, os
sys.path.insert(0, '.')
import torch, bench
from vqnerf_release_amd import _C
from vqnerf_release_amd.geo import train_programs as tp
dev = torch.device('cuda:0')
runner, step = bench.geo_train_setup(dev, 0, 2560)
for phase, mode in enumerate(['f32'] * 8 + ['bf16x3'] * 2 + ['f32'] * 3):
    tp.wgrad_mode(mode)
    _C.KernelClock.reset(True)
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    clk = _C.KernelClock.summary(); _C.KernelClock.reset(False)
    print(phase, mode, f'{dt*1e3:.2f} ms/step', {k: round(v[1] / 5, 2) for k, v in clk.items() if 'wgrad' in k})
