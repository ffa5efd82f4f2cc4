"""Dev probe: time of vqn_brdf_shade_fwd on a 640,000-point view (two material sets + split; one set + 16 probes; no visibility
rows + gamma) -- HIP events around each launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vqnerf_release_amd import _C
from vqnerf_release_amd.decomp.brdf.renderer import gen_light_xyz
dev = torch.device('cuda:0')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 640000
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.rand(*s, device=dev, generator=g)
xyz = torch.nn.functional.normalize(torch.randn(N, 3, device=dev, generator=g), dim=-1) * 0.8
nrm = torch.nn.functional.normalize(xyz + 0.1 * torch.randn(N, 3, device=dev, generator=g), dim=-1)
rayo = torch.tensor([[0, 0, 4.0]], device=dev).repeat(N, 1).contiguous()
lvis = (R(N, 512) < 0.7).float()
lxyz, lareas = gen_light_xyz(16, 32)
lxyz = torch.tensor(lxyz.reshape(-1, 3), dtype=torch.float32, device=dev).contiguous()
lareas = torch.tensor(lareas.reshape(-1), dtype=torch.float32, device=dev).contiguous()
light = R(512, 3)
mats = [(R(N, 3), R(N, 3), R(N, 1) * 0.98 + 0.02) for _ in range(2)]
probes = R(16, 512, 3) * 2


def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(f'N={N}')
print('2 sets + split, lvis      : %.3f ms' % t(lambda: _C.brdf_shade_fwd(xyz, nrm, rayo, lvis, lxyz, lareas, light, mats, want_split=True)))
print('2 sets, lvis (train fwd)  : %.3f ms' % t(lambda: _C.brdf_shade_fwd(xyz, nrm, rayo, lvis, lxyz, lareas, light, mats, raw=True)))
print('1 set, lvis               : %.3f ms' % t(lambda: _C.brdf_shade_fwd(xyz, nrm, rayo, lvis, lxyz, lareas, light, mats[:1])))
print('1 set + 16 probes, lvis   : %.3f ms' % t(lambda: _C.brdf_shade_fwd(xyz, nrm, rayo, lvis, lxyz, lareas, light, mats[:1], probes=probes)))
print('2 sets + split, no lvis, gamma: %.3f ms' % t(lambda: _C.brdf_shade_fwd(xyz, nrm, rayo, None, lxyz, lareas, light, mats, want_split=True,
                                                                             gamma=torch.tensor([1.3, 0.8], device=dev))))
