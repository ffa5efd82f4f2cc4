"""x3 engine: error vs fp64 and launch time of the sdf-only / fine kernels at the headline shape (one box, A/B via VQN_X3_NACC)."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import geo as og
from vqnerf_release_amd import _C
from tests.test_gpu_neus_x3 import _packed
cfg = og.FULL_CFG
p_sdf, p_col, wb_s, d_s, wb_c, d_c = _packed(cfg, 'f32')
_, _, wb_sx, d_sx, wb_cx, d_cx = _packed(cfg, 'x3')
_, _, wb_sh, d_sh, wb_ch, d_ch = _packed(cfg, 'f16s')
rng = np.random.default_rng(11)
n = 4096
pts = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
dirs = rng.normal(size=(n, 3)).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
P64, D64 = torch.tensor(pts, dtype=torch.float64), torch.tensor(dirs, dtype=torch.float64)
p64 = {k: v.double() for k, v in p_sdf.items()}; c64 = {k: v.double() for k, v in p_col.items()}
with torch.no_grad():
    out64 = og.sdf_forward(p64, cfg, P64)
grad64 = og.sdf_gradient(p64, cfg, P64)
with torch.no_grad():
    rgb64 = og.color_forward(c64, cfg, P64, grad64, D64, out64[:, 1:])
Pg, Dg = torch.tensor(pts).cuda(), torch.tensor(dirs).cuda()
res = {'f32': _C.neus_fine_points(d_s, wb_s, d_c, wb_c, pts=Pg, dirs=Dg),
       'x3': _C.neus_fine_points(d_sx, wb_sx, d_cx, wb_cx, pts=Pg, dirs=Dg, mode='x3'),
       'f16s': _C.neus_fine_points(d_sh, wb_sh, d_ch, wb_ch, pts=Pg, dirs=Dg, mode='f16s')}
for m, r in res.items():
    e = [(float((x.double().cpu() - ref).abs().max()), float((x.double().cpu() - ref).abs().mean()), float((x.double().cpu() - ref).mean()))
         for x, ref in zip(r, (out64[:, 0], grad64, rgb64))]
    print(m, 'NACC', os.environ.get('VQN_X3_NACC', '-'), ' sdf max %.2e mean %.2e bias %+.2e | grad max %.2e mean %.2e | rgb max %.2e mean %.2e' %
          (e[0][0], e[0][1], e[0][2], e[1][0], e[1][1], e[2][0], e[2][1]))
# timing at 80,000 rays x 128 samples
B, S = 80000, 128
o = torch.zeros(B, 3).cuda(); o[:, 2] = 4.0
d = torch.nn.functional.normalize(torch.randn(B, 3).cuda() * 0.15 + torch.tensor([0, 0, -1.0]).cuda(), dim=-1)
z = (2.0 + 4.0 * torch.linspace(0, 1, S).cuda())[None].expand(B, S).contiguous()
for mode, (ds_, ws_, dc_, wc_) in {'f32': (d_s, wb_s, d_c, wb_c), 'x3': (d_sx, wb_sx, d_cx, wb_cx), 'f16s': (d_sh, wb_sh, d_ch, wb_ch)}.items():
    for name, fn in (('fine', lambda: _C.neus_fine_points(ds_, ws_, dc_, wc_, rays_o=o, rays_d=d, z=z, mode=mode)),
                     ('sdf', lambda: _C.neus_sdf_points(ds_, ws_, rays_o=o, rays_d=d, z=z, mode=mode))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        print(f'{mode} {name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per {B}x{S}')
