#!/usr/bin/env python
"""A NeuS render through the C ABI ALONE: ctypes on libvqnerf_hip.so, torch only as the owner of device memory, numpy for the
seeded inputs.  Nothing of the Python package is imported (in particular not geo/packing.py): weight packs and descriptors
come from vqn_neus_pack_create / vqn_neus_pack_update, weight-norm from vqn_weight_norm_fwd, and the render is the call
sequence a non-Python host would make (include/vqnerf_hip.h):

    vqn_neus_sdf_points -> 4 x [vqn_neus_upsample -> vqn_neus_sdf_points -> vqn_neus_merge] -> vqn_neus_section_mids
    -> vqn_neus_fine_points -> vqn_neus_composite_fwd

The result is compared with tests/golden/geo_full.npz -- outputs of the REAL reference's NeuSRenderer.render on the same
seeded weights and rays (oracle/gen_golden_geo.py).  Run by tests/test_gpu_cabi.py in a child process; exits non-zero on
any mismatch.  `--f16s` uses the split-precision packs / entry points, `--x3` the exact-split (bf16x3) ones."""
import ctypes
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import geo as og          # seeded weights / rays only (numpy)  # noqa: E402

F16S = '--f16s' in sys.argv
X3 = '--x3' in sys.argv
ENGINE = 2 if X3 else int(F16S)
lib = ctypes.CDLL(os.path.join(ROOT, 'vqnerf_release_amd', 'lib', 'libvqnerf_hip.so'))
lib.vqn_last_error.restype = ctypes.c_char_p
for f in ('vqn_neus_pack_sdf_desc', 'vqn_neus_pack_col_desc', 'vqn_neus_pack_sdf_wbuf', 'vqn_neus_pack_col_wbuf'):
    getattr(lib, f).restype = ctypes.c_void_p
lib.vqn_neus_fine_scratch_bytes.restype = ctypes.c_int64
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev).contiguous()
E = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
STREAM = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ok(rc, what):
    if rc != 0:
        raise SystemExit(f'{what} failed rc={rc}: {lib.vqn_last_error().decode()}')


cfg = og.FULL_CFG
p_sdf, p_col = og.make_sdf_params(cfg, 0), og.make_color_params(cfg, 1)
sdf_dims, col_dims = og.sdf_dims(cfg), og.color_dims(cfg)
n_s, n_c = len(sdf_dims) - 1, len(col_dims) - 1

# ---- effective weights: w = g * v / ||v||_row for every layer of both nets in ONE launch (fields.py:65-66, :139-140) ----
v = [T(p_sdf[f'lin{l}.weight_v']) for l in range(n_s)] + [T(p_col[f'lin{l}.weight_v']) for l in range(n_c)]
g = [T(p_sdf[f'lin{l}.weight_g']).reshape(-1) for l in range(n_s)] + [T(p_col[f'lin{l}.weight_g']).reshape(-1) for l in range(n_c)]
w = [torch.empty_like(t) for t in v]
bias = [T(p_sdf[f'lin{l}.bias']) for l in range(n_s)] + [T(p_col[f'lin{l}.bias']) for l in range(n_c)]
nl = len(v)
arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
rows = (ctypes.c_int32 * nl)(*[t.shape[0] for t in v])
cols = (ctypes.c_int32 * nl)(*[t.shape[1] for t in v])
ok(lib.vqn_weight_norm_fwd(nl, arr(v), arr(g), arr(w), rows, cols, STREAM), 'vqn_weight_norm_fwd')

# ---- packs + descriptors, built in C ----
pack = ctypes.c_void_p()
c = cfg['color']
ok(lib.vqn_neus_pack_create((ctypes.c_int32 * len(sdf_dims))(*sdf_dims), n_s, cfg['sdf']['skip_in'][0], cfg['sdf']['multires'],
                            ctypes.c_float(cfg['sdf']['scale']), 0, c['d_hidden'], c['n_layers'], c['multires_view'], 1, ENGINE,
                            ctypes.byref(pack)), 'vqn_neus_pack_create')
ok(lib.vqn_neus_pack_update(pack, arr(w[:n_s]), arr(bias[:n_s]), arr(w[n_s:]), arr(bias[n_s:]), STREAM), 'vqn_neus_pack_update')
d_s, d_c = ctypes.c_void_p(lib.vqn_neus_pack_sdf_desc(pack)), ctypes.c_void_p(lib.vqn_neus_pack_col_desc(pack))
wb_s, wb_c = ctypes.c_void_p(lib.vqn_neus_pack_sdf_wbuf(pack)), ctypes.c_void_p(lib.vqn_neus_pack_col_wbuf(pack))
sfx = '_x3' if X3 else ('_f16s' if F16S else '')
sdf_points, fine_points = getattr(lib, 'vqn_neus_sdf_points' + sfx), getattr(lib, 'vqn_neus_fine_points' + sfx)

# ---- the render (renderer.py:299-401, perturb 0, white background, cos_anneal_ratio 1) ----
B, n0, radius = 16, 64, 2.0
o, d, near, far = [T(a) for a in og.make_rays(B, 2)]
z = (near + (far - near) * torch.linspace(0.0, 1.0, n0, device=dev)[None, :]).contiguous()
sdf = E(B, n0)
ok(sdf_points(d_s, wb_s, P(o), P(d), P(z), None, ctypes.c_int64(B * n0), n0, P(sdf), STREAM), 'vqn_neus_sdf_points')
u = torch.linspace(0.5 / 16, 1.0 - 0.5 / 16, 16, device=dev).contiguous()
n = n0
for i in range(4):
    z_new = E(B, 16)
    ok(lib.vqn_neus_upsample(P(o), P(d), P(z), P(sdf), ctypes.c_int64(B), n, ctypes.c_float(radius), ctypes.c_float(64.0 * 2 ** i),
                             P(u), 16, P(z_new), STREAM), 'vqn_neus_upsample')
    z_out = E(B, n + 16)
    if i < 3:
        sdf_new, sdf_out = E(B, 16), E(B, n + 16)
        ok(sdf_points(d_s, wb_s, P(o), P(d), P(z_new), None, ctypes.c_int64(B * 16), 16, P(sdf_new), STREAM), 'vqn_neus_sdf_points')
        ok(lib.vqn_neus_merge(P(z), P(sdf), P(z_new), P(sdf_new), ctypes.c_int64(B), n, 16, P(z_out), P(sdf_out), STREAM), 'vqn_neus_merge')
        sdf = sdf_out
    else:
        ok(lib.vqn_neus_merge(P(z), None, P(z_new), None, ctypes.c_int64(B), n, 16, P(z_out), None, STREAM), 'vqn_neus_merge')
    z, n = z_out, n + 16
mid, dists = E(B, n), E(B, n)
ok(lib.vqn_neus_section_mids(P(z), ctypes.c_int64(B), n, ctypes.c_float(2 * radius / n0), None, P(mid), P(dists), STREAM), 'vqn_neus_section_mids')
need = lib.vqn_neus_fine_scratch_bytes(d_s)
scratch = torch.empty(need, dtype=torch.uint8, device=dev)
f_sdf, f_grad, f_rgb = E(B * n), E(B * n, 3), E(B * n, 3)
ok(fine_points(d_s, wb_s, d_c, wb_c, P(o), P(d), P(mid), None, None, ctypes.c_int64(B * n), n, P(scratch), ctypes.c_int64(need),
               P(f_sdf), P(f_grad), P(f_rgb), STREAM), 'vqn_neus_fine_points')
inv_s = T([math.exp(10 * 0.3)])
bg = T([1.0, 1.0, 1.0])
out = dict(color=E(B, 3), weights=E(B, n), cdf=E(B, n), inside=E(B, n), surf=E(B, 3), depth=E(B, 1), weight_sum=E(B, 1),
           weight_max=E(B, 1), gerr=E(B, 2))
ok(lib.vqn_neus_composite_fwd(P(o), P(d), P(mid), P(dists), P(f_sdf), P(f_grad), P(f_rgb), P(inv_s), P(bg), ctypes.c_int64(B), n,
                              ctypes.c_float(radius), ctypes.c_float(1.0), P(out['color']), P(out['weights']), P(out['cdf']),
                              P(out['inside']), P(out['surf']), P(out['depth']), P(out['weight_sum']), P(out['weight_max']),
                              P(out['gerr']), None, STREAM), 'vqn_neus_composite_fwd')
torch.cuda.synchronize()
lib.vqn_neus_pack_destroy(pack)

# ---- against the reference's own outputs ----
gold = np.load(os.path.join(ROOT, 'tests', 'golden', 'geo_full.npz'))
tol = dict(color=1e-3, weights=2e-3, cdf=2e-3, inside=0, surf=2e-3, depth=2e-3, weight_sum=1e-3, weight_max=1e-3)
names = dict(color='color_fine', weights='weights', cdf='cdf_fine', inside='inside_sphere', surf='surf', depth='depth',
             weight_sum='weight_sum', weight_max='weight_max')
for k, t in tol.items():
    ref = gold['render_white_1.0_' + names[k]]
    err = float(np.abs(out[k].cpu().numpy().reshape(ref.shape) - ref).max())
    assert err <= t, (k, err, t)
ge = out['gerr'].sum(0)
assert abs(float(ge[0] / (ge[1] + 1e-5)) - float(gold['render_white_1.0_gradient_error'])) <= 1e-4
np.testing.assert_allclose(f_grad.cpu().numpy().reshape(B, n, 3), gold['render_white_1.0_gradients'], rtol=0, atol=2e-3)
psnr = -10 * np.log10(np.mean((out['color'].cpu().numpy() - gold['render_white_1.0_color_fine']) ** 2) + 1e-20)
assert psnr > 70, psnr
mods = [m for m in sys.modules if m.startswith('vqnerf_release_amd')]
assert not mods, mods                                   # the Python package was never imported
print(f'C-ABI-only render ({"exact-split bf16x3" if X3 else ("split-precision" if F16S else "f32")}): PSNR vs the reference = {psnr:.1f} dB; imported package modules: {mods}')
