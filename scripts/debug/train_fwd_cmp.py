"""Fused training forward (vqn_neus_train_fwd) against the interpreted prog_fwd: every saved tensor, then step times."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, bench
from vqnerf_release_amd import _C
dev = torch.device('cuda:0')
runner, step = bench.geo_train_setup(dev, 0, 2560, graph=False)
ren = runner.renderer
eng = ren._train_engine(runner.sdf_network, runner.color_network)
print('fused_forward:', eng.fused_forward())
P = 4096 + 17
g = torch.Generator(device='cuda').manual_seed(3)
x = (torch.rand(P, 3, device=dev, generator=g) * 2 - 1)
d = torch.nn.functional.normalize(torch.randn(P, 3, device=dev, generator=g), dim=-1)
sdf_l = [getattr(runner.sdf_network, 'lin%d' % l) for l in range(runner.sdf_network.num_layers - 1)]
col_l = [getattr(runner.color_network, 'lin%d' % l) for l in range(runner.color_network.num_layers - 1)]
with torch.no_grad():
    W, b = [m.effective_weight().float() for m in sdf_l], [m.bias.float() for m in sdf_l]
    Wc, bc = [m.effective_weight().float() for m in col_l], [m.bias.float() for m in col_l]
    wbuf, descs, flat = eng.pack(W, b, Wc, bc, want_flat=True)
    Ta, Tb = eng.alloc_tensors(P, dev), eng.alloc_tensors(P, dev)
    for T in (Ta, Tb):
        T['X'].copy_(x); T['DIRS'].copy_(d)
        for k in T:
            if k not in ('X', 'DIRS', 'ONES'): T[k].fill_(float('nan'))
    eng.run('prog_fwd', descs, wbuf, Ta, P)
    eng.run_fused_forward(flat, Tb, P)
    torch.cuda.synchronize()
    names = ['SDF', 'N', 'RGB', 'E', 'OUTF', 'EXTR'] + ['U%d' % (l + 1) for l in range(eng.nL)] + ['GH%d' % l for l in range(eng.nL)] + ['C%d' % (l + 1) for l in range(eng.nC)]
    width = {'E': eng.E, 'OUTF': eng.F, 'EXTR': eng.X}
    for l in range(eng.nL):
        width['U%d' % (l + 1)] = width['GH%d' % l] = eng.out[l]
    for l in range(eng.nC):
        width['C%d' % (l + 1)] = eng.cout[l]
    worst = 0
    for n in names:
        a, c = Ta[n], Tb[n]
        if a.dim() == 4:                       # [tiles, ft, 32 feats, 32 points] -> [points, feats], valid part
            a = a.permute(0, 3, 1, 2).reshape(a.shape[0] * 32, -1)[:P, :width[n]]
            c = c.permute(0, 3, 1, 2).reshape(c.shape[0] * 32, -1)[:P, :width[n]]
        nan = int(torch.isnan(c).sum())
        err = float((a - c).abs().max()); sc = float(a.abs().max())
        worst = max(worst, err / max(sc, 1e-12))
        print(f'{n:6s} max|prog| {sc:.4e}  max diff {err:.3e}  rel {err/max(sc,1e-12):.2e}  nan {nan}')
    print('worst rel', worst)
    Tc = eng.alloc_tensors(P, dev)
    Tc['X'].copy_(x); Tc['DIRS'].copy_(d)
    for k in Tc:
        if k not in ('X', 'DIRS', 'ONES'): Tc[k].fill_(float('nan'))
    eng.run_fused_forward_x3(W, b, Wc, bc, Tc, P)
    torch.cuda.synchronize()
    worst = 0
    for n in names:
        a, c = Ta[n], Tc[n]
        if a.dim() == 4:
            a = a.permute(0, 3, 1, 2).reshape(a.shape[0] * 32, -1)[:P, :width[n]]
            c = c.permute(0, 3, 1, 2).reshape(c.shape[0] * 32, -1)[:P, :width[n]]
        nan = int(torch.isnan(c).sum())
        err = float((a - c).abs().max()); sc = float(a.abs().max())
        worst = max(worst, err / max(sc, 1e-12))
        print(f'x3 {n:6s} max|prog| {sc:.4e}  max diff {err:.3e}  rel {err/max(sc,1e-12):.2e}  nan {nan}')
    print('x3 worst rel', worst)
for mode in ('prog', 'fused', 'x3'):
    os.environ['VQN_TRAIN_FWD'] = mode
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); print(f'VQN_TRAIN_FWD={mode}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step (eager)')
    _C.KernelClock.reset(True)
    step(); torch.cuda.synchronize()
    summ = _C.KernelClock.summary()
    _C.KernelClock.reset(False)
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])[:7]:
        print('   ', k, v)
