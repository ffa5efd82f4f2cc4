#!/usr/bin/env python
"""The reflectance path through the C ABI ALONE (ctypes on libvqnerf_hip.so, torch only as the owner of device memory): layer
programs and weight packs from vqn_chain_pack_create / _update -- not from decomp/packing.py -- then the call sequence of
vq_nfr.Model.call in inference mode (vq_nfr.py:534-692):

    vqn_mlp_chain_fwd (encoder + continuous heads, one program) -> vqn_vq_quantize_rows -> vqn_mlp_chain_fwd (VQ heads)
    -> vqn_brdf_shade_fwd (both material sets + diffuse / specular split)

compared with oracle/decomp.py (model_call, mode 'vali') on the same seeded weights and points, at the tolerances of
tests/test_gpu_decomp.py.  Run by tests/test_gpu_cabi.py in a child process; exits non-zero on any mismatch."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import decomp as od          # the checker (CPU)  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, 'vqnerf_release_amd', 'lib', 'libvqnerf_hip.so'))
lib.vqn_last_error.restype = ctypes.c_char_p
lib.vqn_chain_pack_desc.restype = lib.vqn_chain_pack_wbuf.restype = ctypes.c_void_p
dev = torch.device('cuda:0')
P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev).contiguous()
E = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
STREAM = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ACT = {None: 0, 'relu': 1, 'sigmoid': 3}


def ok(rc, what):
    if rc != 0:
        raise SystemExit(f'{what} failed rc={rc}: {lib.vqn_last_error().decode()}')


class Stack(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('n_layers', ctypes.c_int32), ('widths', ctypes.c_int32 * 8), ('acts', ctypes.c_int32 * 8),
                ('skip_at', ctypes.c_int32), ('input', ctypes.c_int32), ('out_slot', ctypes.c_int32)]


def stack(kind, spec, input=-1, out_slot=-1):
    s = Stack()
    s.kind, s.n_layers, s.input, s.out_slot = kind, len(spec['widths']), input, out_slot
    s.skip_at = spec['skip_at'][0] if spec['skip_at'] else -1
    for i, (w, a) in enumerate(zip(spec['widths'], spec['act'])):
        s.widths[i], s.acts[i] = w, ACT[a]
    return s


def program(in_mode, in_feats, n_freqs, stacks, nets, p):
    pack = ctypes.c_void_p()
    arr = (Stack * len(stacks))(*stacks)
    ok(lib.vqn_chain_pack_create(in_mode, in_feats, n_freqs, len(stacks), arr, ctypes.byref(pack)), 'vqn_chain_pack_create')
    ws = [T(W) for n in nets for W, _ in p[n]]
    bs = [T(b) for n in nets for _, b in p[n]]
    assert lib.vqn_chain_pack_n_weights(pack) == len(ws)
    ptrs = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    ok(lib.vqn_chain_pack_update(pack, ptrs(ws), ptrs(bs), STREAM), 'vqn_chain_pack_update')
    return pack, (ws, bs)


def run(pack, x, widths):
    N = x.shape[0]
    outs = [E(N, w) for w in widths] + [None] * (4 - len(widths))
    lds = list(widths) + [0] * (4 - len(widths))
    ok(lib.vqn_mlp_chain_fwd(ctypes.c_void_p(lib.vqn_chain_pack_desc(pack)), ctypes.c_void_p(lib.vqn_chain_pack_wbuf(pack)), P(x), ctypes.c_int64(N),
                             P(outs[0]), lds[0], P(outs[1]), lds[1], P(outs[2]), lds[2], P(outs[3]), lds[3], STREAM), 'vqn_mlp_chain_fwd')
    return outs[:len(widths)]


K, N = 15, 3000
p, specs = od.make_model_params(seed=0, K=K)
pts = od.make_points(N, seed=3)
xyz, normal, rayo, lvis = T(pts['xyz']), T(pts['normal']), T(pts['rayo']), T(pts['lvis'])

# encoder + continuous-branch heads: ONE program, z stays in LDS for the heads
enc_main, keep1 = program(1, 63, 10, [stack(0, specs['fine_enc']), stack(0, specs['bottleneck'], input=0, out_slot=0)] +
                          [stack(1, specs[n], input=1, out_slot=i + 1) for i, n in enumerate(('diff_main', 'spec_main', 'rough_main'))],
                          ['fine_enc', 'bottleneck', 'diff_main', 'spec_main', 'rough_main'], p)
z, base, ks, rough = run(enc_main, xyz, [256, 3, 1, 1])
spec, albedo = (ks * base).contiguous(), ((1 - ks) * base).contiguous()

# codebook (vq_nfr.py:761-769) and the fused quantiser
cb = T(p['codebook_raw']).clamp(0.0, 1.0)
cb = (cb * torch.rsqrt(torch.clamp((cb * cb).sum(0, keepdim=True), min=1e-6))).contiguous()
idx = torch.empty(N, dtype=torch.int64, device=dev)
ste, loss, counts, ws = E(N, 256), E(1), E(K), E(4096)
ok(lib.vqn_vq_quantize_rows(P(z), ctypes.c_int64(N), 256, P(cb), K, None, ctypes.c_float(1e-6), ctypes.c_float(1.0 / (N * 256)), P(ws), P(idx),
                            P(ste), P(loss), P(counts), STREAM), 'vqn_vq_quantize_rows')
vq_heads, keep2 = program(0, 256, 0, [stack(1, specs[n], out_slot=i) for i, n in enumerate(('diff_vq', 'spec_vq', 'rough_vq'))],
                          ['diff_vq', 'spec_vq', 'rough_vq'], p)
vq_albedo, vq_spec, vq_rough = run(vq_heads, ste, [3, 3, 1])

lxyz, lareas = od.gen_light_xyz(16, 32)
light = T(p['light']).clamp(min=0).reshape(-1, 3).contiguous()
rgb0, rgb1, rd, rs, nout = E(N, 3), E(N, 3), E(N, 3), E(N, 3), E(N, 3)
ok(lib.vqn_brdf_shade_fwd(P(xyz), P(normal), P(rayo), P(lvis), P(T(lxyz.reshape(-1, 3))), P(T(lareas.reshape(-1))), P(light), ctypes.c_int64(N), 512, 2,
                          P(albedo), P(spec), P(rough), P(vq_albedo), P(vq_spec), P(vq_rough), None, P(nout), P(rgb0), P(rgb1), P(rd), P(rs), 0,
                          None, 0, None, STREAM), 'vqn_brdf_shade_fwd')
# ---- the same front end as ONE launch: encoder + heads -> VQ step on z in LDS -> VQ heads (vqn_mlp_chain_vq_fwd) ----
frags = E(16 * 64 * 4 + 16)
ok(lib.vqn_vq_codebook_frags(P(cb), 256, K, P(frags), STREAM), 'vqn_vq_codebook_frags')
f_base, f_ks, f_rough, f_va, f_vs, f_vr = E(N, 3), E(N, 1), E(N, 1), E(N, 3), E(N, 3), E(N, 1)
f_idx = torch.empty(N, dtype=torch.int64, device=dev)
f_loss, f_counts = E(1), E(K)
tab = lambda ts: (ctypes.c_void_p * 4)(*[(0 if t is None else t.data_ptr()) for t in ts])
i4 = lambda v: (ctypes.c_int32 * 4)(*v)
ok(lib.vqn_mlp_chain_vq_fwd(ctypes.c_void_p(lib.vqn_chain_pack_desc(enc_main)), ctypes.c_void_p(lib.vqn_chain_pack_wbuf(enc_main)),
                            ctypes.c_void_p(lib.vqn_chain_pack_desc(vq_heads)), ctypes.c_void_p(lib.vqn_chain_pack_wbuf(vq_heads)),
                            P(xyz), ctypes.c_int64(N), tab([None, f_base, f_ks, f_rough]), i4([0, 3, 1, 1]),
                            tab([f_va, f_vs, f_vr, None]), i4([3, 3, 1, 0]), P(frags), K, ctypes.c_float(1e-6),
                            ctypes.c_float(1.0 / (N * 256)), P(f_idx), None, P(f_loss), P(f_counts), P(ws), STREAM), 'vqn_mlp_chain_vq_fwd')
torch.cuda.synchronize()
for a, b, what in ((f_base, base, 'basecolor'), (f_ks, ks, 'ks'), (f_rough, rough, 'rough'), (f_va, vq_albedo, 'vq_albedo'),
                   (f_vs, vq_spec, 'vq_spec'), (f_vr, vq_rough, 'vq_rough'), (f_idx, idx, 'indices'), (f_counts, counts, 'counts')):
    assert torch.equal(a, b), 'one-launch front differs from the separate launches: ' + what
np.testing.assert_allclose(float(f_loss), float(loss), rtol=2e-6)
for h in (enc_main, vq_heads):
    lib.vqn_chain_pack_destroy(h)

# ---- against the oracle ----
pt = {k: ([(od.T(W), od.T(b)) for W, b in v] if isinstance(v, list) else od.T(v)) for k, v in p.items()}
ob = {k: od.T(v) for k, v in pts.items()}
want = od.model_call(pt, specs, ob, od.T(lxyz), od.T(lareas), od.EMA(0.999, (K,)), od.EMA(0.999, (256, K)), mode='vali')
chk = lambda got, ref, tol, what: np.testing.assert_allclose(got.cpu().numpy(), ref.numpy(), rtol=0, atol=tol, err_msg=what)
chk(z, want['z_enc'], 3e-6, 'z')
chk(albedo, want['albedo'], 5e-6, 'albedo'); chk(spec, want['spec'], 5e-6, 'spec'); chk(rough, want['rough'], 5e-6, 'rough')
np.testing.assert_array_equal(idx.cpu().numpy(), want['vq']['encoding_indices'].numpy())             # VQ indices exact
chk(ste, want['z_vq'], 1e-6, 'z_vq')
chk(vq_albedo, want['vq_albedo'], 5e-6, 'vq_albedo')
chk(rgb0, want['rgb'], 2e-5, 'rgb'); chk(rgb1, want['vq_rgb'], 2e-5, 'vq_rgb'); chk(rd, want['rgb_diff'], 2e-5, 'rgb_diff')
np.testing.assert_allclose(float(loss) * 0.1, float(want['vq']['loss']), rtol=1e-4)
assert float(counts.sum()) == N
mods = [m for m in sys.modules if m.startswith('vqnerf_release_amd')]
assert not mods, mods
psnr = -10 * np.log10(np.mean((rgb0.cpu().numpy() - want['rgb'].numpy()) ** 2) + 1e-20)
print(f'C-ABI-only reflectance call: PSNR(rgb) vs the oracle = {psnr:.1f} dB, VQ indices exact over {N} rows; imported package modules: {mods}')
