"""K = 64 assignment: how many rows / 16-row groups does the prefilter leave undecided (more than one candidate within its margin)?
Statistics in torch with the kernel's margin formula (csrc/vq.hip: mrow), for the bench's three row sets."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
from vqnerf_release_amd import _C
dev = torch.device('cuda:0')
N, D, K = 1 << 20, 256, 64
g = torch.Generator(device='cuda').manual_seed(0)
def stats(x, C, label):
    x2 = (x * x).sum(1)
    c2 = (C * C).sum(0)
    d = c2[None, :] - 2.0 * (x @ C)
    sx, sc = x2.sqrt(), c2.max().sqrt()
    m = 1.72e-4 * sx * sc + 1.2e-6 * (x2 + c2.max()) + 1e-9 * (sx + sc)
    for scale in (1.0, 0.4):
        cand = (d <= (d.min(1, keepdim=True).values + scale * m[:, None])).sum(1)
        multi = cand > 1
        grp = multi.reshape(-1, 16)
        gm = grp.any(1)
        over4 = (cand.reshape(-1, 16).max(1).values > 4)
        print(f'{label:28s} margin x{scale}: rows multi {multi.float().mean()*100:6.2f} %  groups with a multi row {gm.float().mean()*100:6.2f} %  '
              f'groups > 4 cand {over4.float().mean()*100:6.2f} %  mean cand of multi rows {cand[multi].float().mean():.2f}  mean max-cand of multi groups {cand.reshape(-1,16).max(1).values[gm].float().mean():.2f}')
cb = torch.nn.functional.normalize(torch.rand(D, K, device=dev, generator=g), dim=0).contiguous()
x_rand = torch.nn.functional.normalize(torch.rand(N, D, device=dev, generator=g), dim=1).contiguous()
stats(x_rand, cb, 'uniform rows')
idx = torch.randint(0, K, (N,), device=dev, generator=g)
x_near = torch.nn.functional.normalize(cb.t()[idx] + 0.05 * torch.randn(N, D, device=dev, generator=g), dim=1).contiguous()
stats(x_near, cb, 'rows near codes')
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict
model = get_model_class('vq_nfr')(config_from_dict(bench.DECOMP_INI))
model.build_nets(device=dev, seed=0).to(dev)
with torch.no_grad():
    xyz_e = torch.nn.functional.normalize(torch.randn(N, 3, device=dev), dim=-1) * (0.5 + 0.5 * torch.rand(N, 1, device=dev))
    enc_rows = _C.l2_normalize_rows(model._pred_enc_at(xyz_e).contiguous())
    C_enc = enc_rows[torch.randperm(N, device=dev)[:K]].t().contiguous()
stats(enc_rows, C_enc, 'encoder rows')
for lab, x, C in (('uniform', x_rand, cb), ('near', x_near, cb), ('encoder', enc_rows, C_enc)):
    for _ in range(3): _C.vq_assign(x, C, want_quant=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): _C.vq_assign(x, C, want_quant=False)
    e1.record(); torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 20 * 1e-3
    print(f'{lab:10s} {dt*1e3:.4f} ms  {(N*(4*D+8)+4*D*K)/dt/1e12:.3f} TB/s')
