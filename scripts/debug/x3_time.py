"""Launch time of the x3 sdf-only / fine kernels at 80,000 rays x 128 samples; VQN_LIB picks a diagnostic build of the library."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import geo as og
from vqnerf_release_amd import _C
from tests.test_gpu_neus_x3 import _packed
cfg = og.FULL_CFG
modes = sys.argv[1:] or ['x3']
B, S = 80000, 128
o = torch.zeros(B, 3).cuda(); o[:, 2] = 4.0
d = torch.nn.functional.normalize(torch.randn(B, 3).cuda() * 0.15 + torch.tensor([0, 0, -1.0]).cuda(), dim=-1)
z = (2.0 + 4.0 * torch.linspace(0, 1, S).cuda())[None].expand(B, S).contiguous()
for mode in modes:
    _, _, ws_, ds_, wc_, dc_ = _packed(cfg, mode)
    for name, fn in (('fine', lambda: _C.neus_fine_points(ds_, ws_, dc_, wc_, rays_o=o, rays_d=d, z=z, mode=mode)),
                     ('sdf', lambda: _C.neus_sdf_points(ds_, ws_, rays_o=o, rays_d=d, z=z, mode=mode))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        print(f'{os.environ.get("VQN_LIB", "default")} NACC={os.environ.get("VQN_X3_NACC", "-")} {mode} {name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms')
