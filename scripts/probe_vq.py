"""Timing probe: vqn_vq_assign / vqn_vq_ema_stats at 1 M rows (HBM roofline rows K8 / K9)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vqnerf_release_amd import _C

def t_gpu(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

N, D = (int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20), 256
g = torch.Generator(device='cuda'); g.manual_seed(0)
x = torch.rand((N, D), device='cuda', generator=g)
for K in (15, 32, 64):
    C = torch.rand((D, K), device='cuda', generator=g)
    idx, _, _ = _C.vq_assign(x, C, want_quant=False)
    ta = t_gpu(lambda: _C.vq_assign(x, C, want_quant=False))
    te = t_gpu(lambda: _C.vq_ema_stats(x, idx, K))
    ba = N * (4 * D + 8) + 4 * D * K
    be = N * (4 * D + 8) + 4 * K * (D + 1)
    print(f'K={K}: assign {ta*1e3:.3f} ms {ba/ta/1e9:.0f} GB/s   ema {te*1e3:.3f} ms {be/te/1e9:.0f} GB/s', flush=True)

# the fused quantiser (l2-normalise + assign + straight-through + commitment + counts) on encoder-like rows
z = torch.sigmoid(torch.randn((N, D), device='cuda', generator=g))
for K in (15, 32, 64):
    C = torch.rand((D, K), device='cuda', generator=g)
    C = C / C.norm(dim=0, keepdim=True)
    tq = t_gpu(lambda: _C.vq_quantize_rows(z, C))
    bq = N * (8 * D + 8) + 4 * D * K
    print(f'K={K}: quantize_rows {tq*1e3:.3f} ms {bq/tq/1e9:.0f} GB/s (variant {_C.vq_assign_variant(D, K)})', flush=True)
