"""K = 64 index-only assignment: random rows (the bench's) against rows near the codes (what a trained encoder produces): how much of the
time is the exact re-evaluation of near-ties?"""
import sys, time
sys.path.insert(0, '.')
import torch
from vqnerf_release_amd import _C
dev = torch.device('cuda:0')
N, D, K = 1 << 20, 256, 64
g = torch.Generator(device='cuda').manual_seed(0)
cb = torch.nn.functional.normalize(torch.rand(D, K, device=dev, generator=g), dim=0).contiguous()
def timed(x, label):
    for _ in range(3): _C.vq_assign(x, cb, want_quant=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): _C.vq_assign(x, cb, want_quant=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f'{label:40s} {dt*1e3:.3f} ms  {N*D*4/dt/1e12:.2f} TB/s')
x_rand = torch.nn.functional.normalize(torch.rand(N, D, device=dev, generator=g), dim=1).contiguous()
idx = torch.randint(0, K, (N,), device=dev, generator=g)
x_near = torch.nn.functional.normalize(cb.t()[idx] + 0.05 * torch.randn(N, D, device=dev, generator=g), dim=1).contiguous()
x_randn = torch.nn.functional.normalize(torch.randn(N, D, device=dev, generator=g), dim=1).contiguous()
timed(x_rand, 'uniform[0,1) rows, normalised (bench)')
timed(x_randn, 'gaussian rows, normalised')
timed(x_near, 'rows = code + 5 % noise')
