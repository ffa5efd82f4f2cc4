"""Dev probe: in-kernel phase breakdown of the fused NeuS kernels from the DIAGNOSTIC build (make -C vqnerf_release_amd/csrc stamps).
Usage:  VQN_LIB=vqnerf_release_amd/lib/libvqnerf_hip_stamps.so python scripts/probe_stamps.py [rays]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from vqnerf_release_amd import _C
from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork
from vqnerf_release_amd.geo.models.renderer import NeuSRenderer

assert 'stamps' in _C.LIB_PATH, 'set VQN_LIB to the stamps build'
torch.manual_seed(0)
sdf, col, var = SDFNetwork(**bench.FULL['sdf']).cuda(), RenderingNetwork(**bench.FULL['color']).cuda(), SingleVarianceNetwork(0.3).cuda()
ren = NeuSRenderer(None, sdf, var, col, **bench.FULL['renderer'])
wb_s, d_s, wb_c, d_c = ren._packs()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
o_np, d_np = bench.image_rays(np.arange((B + 799) // 800))
o, d = torch.tensor(o_np[:B]).cuda(), torch.tensor(d_np[:B]).cuda()
near, far = torch.full((B, 1), 2.0).cuda(), torch.full((B, 1), 6.0).cuda()
lib = _C.lib()
names = ['tile set-up', 'GEMM main loops (hidden layers, incl. operand waits)', 'epilogues (hidden layers)', 'barrier waits (hidden layers)',
         'everything else (last layer, reverse sweep, colour net, outputs)', 'total', 'workgroups']
for S, fine in ((64, False), (128, True)):
    z = (near + (far - near) * torch.linspace(0, 1, S, device='cuda')[None, :]).contiguous()
    f = (lambda: _C.neus_fine_points(d_s, wb_s, d_c, wb_c, rays_o=o, rays_d=d, z=z)) if fine else (lambda: _C.neus_sdf_points(d_s, wb_s, rays_o=o, rays_d=d, z=z))
    f(); torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 8)()
    lib.vqn_debug_read_stamps(buf, 1)
    f(); torch.cuda.synchronize()
    lib.vqn_debug_read_stamps(buf, 1)
    v = [int(x) for x in buf]
    print('fine' if fine else 'sdf ', f'P={B*S}  wave-0 cycles per workgroup: total {v[5] / max(v[6], 1):.3e} over {v[6]} workgroups')
    for i in range(5):
        print(f'    {names[i]:72s} {100.0 * v[i] / max(v[5], 1):5.1f} %')

# ---- split-precision kernels (16 phases) ----
if hasattr(lib, 'vqn_debug_read_stamps16'):
    names16 = ['set-up (points, posenc)', 'forward K loops', 'forward epilogues (softplus, act\', split, stash)', 'forward barrier waits',
               'sdf row + feature layer + sdf out', 'G_pre', 'reverse K loops (+ wTE)', 'reverse epilogues', 'reverse barrier waits',
               'embedding chain rule', 'colour set-up (extras, features from stash)', 'colour layers', 'colour out / tile turn-around']
    wb_s16, d_s16 = sdf.packs(max_tiles=col.max_tiles(), mode='f16s')
    wb_c16, d_c16 = col.packs(feat_tiles=sdf.plan(mode='f16s').tiles[-1], mode='f16s')
    for S, fine in ((64, False), (128, True)):
        z = (near + (far - near) * torch.linspace(0, 1, S, device='cuda')[None, :]).contiguous()
        f = (lambda: _C.neus_fine_points(d_s16, wb_s16, d_c16, wb_c16, rays_o=o, rays_d=d, z=z, mode='f16s')) if fine \
            else (lambda: _C.neus_sdf_points(d_s16, wb_s16, rays_o=o, rays_d=d, z=z, mode='f16s'))
        f(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        lib.vqn_debug_read_stamps16(buf, 1)
        f(); torch.cuda.synchronize()
        lib.vqn_debug_read_stamps16(buf, 1)
        v = [int(x) for x in buf]
        print('f16s fine' if fine else 'f16s sdf ', f'P={B*S}  wave-0 cycles per workgroup: total {v[14] / max(v[15], 1):.3e} over {v[15]} workgroups')
        for i in range(13):
            print(f'    {names16[i]:72s} {100.0 * v[i] / max(v[14], 1):5.1f} %')

# ---- two-image f32 kernels (16 phases) ----
if hasattr(lib, 'vqn_debug_read_stamps2'):
    for S, fine in ((64, False), (128, True)):
        z = (near + (far - near) * torch.linspace(0, 1, S, device='cuda')[None, :]).contiguous()
        f = (lambda: _C.neus_fine_points(d_s, wb_s, d_c, wb_c, rays_o=o, rays_d=d, z=z)) if fine else (lambda: _C.neus_sdf_points(d_s, wb_s, rays_o=o, rays_d=d, z=z))
        f(); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        lib.vqn_debug_read_stamps2(buf, 1)
        f(); torch.cuda.synchronize()
        lib.vqn_debug_read_stamps2(buf, 1)
        v = [int(x) for x in buf]
        print('f32 two-image fine' if fine else 'f32 two-image sdf ', f'P={B*S}  wave-0 cycles per workgroup: total {v[14] / max(v[15], 1):.3e} over {v[15]} workgroups')
        for i in range(13):
            print(f'    {names16[i]:72s} {100.0 * v[i] / max(v[14], 1):5.1f} %')
