"""Dev probe: time the fused NeuS kernels alone (not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import geo as og
from vqnerf_release_amd import _C
from vqnerf_release_amd.geo import packing as pk

cfg = og.FULL_CFG
c, cc = cfg['sdf'], cfg['color']
sp = pk.SdfPackPlan(og.sdf_dims(cfg), c['skip_in'], c['multires'], c['scale'], max_tiles=8)
cp = pk.ColPackPlan(cc['d_feature'], cc['mode'], cc['d_hidden'], cc['n_layers'], cc['d_out'], cc['multires_view'], cc['squeeze_out'], feat_tiles=sp.tiles[-1])
p_sdf = og.to_torch(og.make_sdf_params(cfg, 0)); p_col = og.to_torch(og.make_color_params(cfg, 1))
Ws = [og.wn_weight(p_sdf, l).cuda() for l in range(sp.n_lin)]; bs = [p_sdf[f'lin{l}.bias'].cuda() for l in range(sp.n_lin)]
Wc = [og.wn_weight(p_col, l).cuda() for l in range(cp.n_lin)]; bc = [p_col[f'lin{l}.bias'].cuda() for l in range(cp.n_lin)]
wb_s, d_s = sp.pack(Ws, bs); wb_c, d_c = cp.pack(Wc, bc)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
o, d, near, far = map(lambda a: torch.tensor(a).cuda(), og.make_rays(B, 2))
for S, fine in ((64, False), (128, True)):
    z = (near + (far - near) * torch.linspace(0, 1, S, device='cuda')[None, :]).contiguous()
    f = (lambda: _C.neus_fine_points(d_s, wb_s, d_c, wb_c, rays_o=o, rays_d=d, z=z)) if fine else (lambda: _C.neus_sdf_points(d_s, wb_s, rays_o=o, rays_d=d, z=z))
    for _ in range(2): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    n = 5; e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    P = B * S
    macs = (2 * 524544 + 271360) if fine else 524544
    print(f"{'fine' if fine else 'sdf '} P={P} {ms:.3f} ms  {P/ms*1e3/1e6:.2f} Mpts/s  algorithmic {2*macs*P/ms*1e3/1e12:.1f} TFLOP/s")
