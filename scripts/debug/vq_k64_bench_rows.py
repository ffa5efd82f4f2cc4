"""The bench's K = 64 assignment / quantise legs alone (bench.py: extra.vq_assign_k64 incl. rows_near_codes / rows_encoder_outputs, extra.vq_quantize_rows_k64),
seeded, so that two library builds (VQN_LIB) see the same rows and codes."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch, bench
from vqnerf_release_amd import _C
from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict
from vqnerf_release_amd.decomp.nerfactor.networks.vq_layers import VectorQuantizerEMA
dev = torch.device('cuda:0')
torch.manual_seed(0)
rng = np.random.default_rng(1)
Nv, D, K64 = 1 << 20, 256, 64
cb64 = rng.uniform(0, 1, (K64, 256)).astype(np.float32)
C64 = torch.tensor((cb64 / np.linalg.norm(cb64, axis=1, keepdims=True)).T.copy(), device=dev)
x = torch.rand(Nv, D, device=dev); x = x / x.norm(dim=1, keepdim=True)
by = Nv * (4 * D + 8) + 4 * D * K64
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
res = {}
res['uniform'] = t(lambda: _C.vq_assign(x, C64, want_quant=False))
near = torch.nn.functional.normalize(C64.t()[torch.randint(0, K64, (Nv,), device=dev)] + 0.05 * torch.randn(Nv, D, device=dev), dim=1).contiguous()
res['near'] = t(lambda: _C.vq_assign(near, C64, want_quant=False))
model = get_model_class('vq_nfr')(config_from_dict(bench.DECOMP_INI)); model.build_nets(device=dev, seed=0).to(dev)
for draw in range(3):
    with torch.no_grad():
        xyz_e = torch.nn.functional.normalize(torch.randn(Nv, 3, device=dev), dim=-1) * (0.5 + 0.5 * torch.rand(Nv, 1, device=dev))
        enc_rows = _C.l2_normalize_rows(model._pred_enc_at(xyz_e).contiguous())
        C_enc = enc_rows[torch.randperm(Nv, device=dev)[:K64]].t().contiguous()
    res[f'encoder draw {draw}'] = t(lambda: _C.vq_assign(enc_rows, C_enc, want_quant=False))
xr = torch.rand(Nv, D, device=dev)
vql = VectorQuantizerEMA(embedding_dim=D, num_embeddings=K64, commitment_cost=0.1, seed=0).to(dev)
with torch.no_grad():
    tq = t(lambda: vql.infer_from_raw(xr, C64)['quantize'], 10)
for k, v in res.items():
    print(f'{k:18s} {v*1e3:.4f} ms  frac {by / v / 1e9 / 8000:.3f}')
print(f'quantize_rows      {tq*1e3:.4f} ms  frac {(Nv * (8 * D + 8) + 4 * D * K64) / tq / 1e9 / 8000:.3f}')
