"""Dev probe: the split-precision NeuS kernels alone (SDF-only at 64 samples, fine at 128) next to the f32 ones."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vqnerf_release_amd import _C
from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork
torch.manual_seed(0)
sdf, col = SDFNetwork(**bench.FULL['sdf']).cuda(), RenderingNetwork(**bench.FULL['color']).cuda()
B = int(os.environ.get('PROBE_B', 40960))
o_np, d_np = bench.image_rays(np.arange((B + 799) // 800))
o, d = torch.tensor(o_np[:B]).cuda(), torch.tensor(d_np[:B]).cuda()
near, far = torch.full((B, 1), 2.0).cuda(), torch.full((B, 1), 6.0).cuda()
for mode in sys.argv[1:] or ['f32', 'f16s']:
    wb_s, d_s = sdf.packs(max_tiles=col.max_tiles(), mode=mode)
    wb_c, d_c = col.packs(feat_tiles=sdf.plan(mode=mode).tiles[-1], mode=mode)
    out = []
    for S, fine in ((64, False), (128, True)):
        z = (near + (far - near) * torch.linspace(0, 1, S, device='cuda')[None, :]).contiguous()
        f = (lambda: _C.neus_fine_points(d_s, wb_s, d_c, wb_c, rays_o=o, rays_d=d, z=z, mode=mode)) if fine else (lambda: _C.neus_sdf_points(d_s, wb_s, rays_o=o, rays_d=d, z=z, mode=mode))
        for _ in range(2): f()
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 5; e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        macs = (2 * 524544 + 271360) if fine else 524544
        out.append(f"{'fine' if fine else 'sdf'} {ms:.2f} ms ({2*macs*B*S/ms*1e3/1e12:.0f} TF)")
    print(os.environ.get('VQN_LIB', 'default').split('/')[-1], mode, '  '.join(out), flush=True)
