"""Dev probe: one 256 x 256 weight-gradient launch over 327,680 points, f32 / bf16x3 contraction kernels."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqnerf_release_amd import _C
lib = _C.lib()
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 10240
A = torch.randn((nt, 8, 32, 32), device='cuda')
B = torch.randn((nt, 8, 32, 32), device='cuda')
ws = torch.empty((256, 256, 256), device='cuda')
for entry in ('vqn_wgrad_partials', 'vqn_wgrad_partials_x3'):
    f = lambda: getattr(lib, entry)(_C._ptr(A), 8, 0, 8, _C._ptr(B), 8, 0, 8, ctypes.c_int64(nt), 256, _C._ptr(ws), None, _C._stream())
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'{entry:28s} {ms*1e3:7.1f} us   {2*256*256*nt*32/ms/1e9:6.1f} TFLOP/s   operands {2*nt*8*4096/ms/1e6:6.0f} GB/s')
