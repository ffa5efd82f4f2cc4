"""Fused training backward (vqn_neus_train_bwd) against the interpreted prog_cbwd + prog_sbwd: every output tensor, then step times."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, bench
from vqnerf_release_amd import _C
dev = torch.device('cuda:0')
runner, step = bench.geo_train_setup(dev, 0, 2560, graph=False)
ren = runner.renderer
eng = ren._train_engine(runner.sdf_network, runner.color_network)
print('fused_backward:', eng.fused_backward())
for P in (4096 + 17, 64):
    g = torch.Generator(device='cuda').manual_seed(3)
    x = (torch.rand(P, 3, device=dev, generator=g) * 2 - 1)
    d = torch.nn.functional.normalize(torch.randn(P, 3, device=dev, generator=g), dim=-1)
    g_rgb = torch.randn(P, 3, device=dev, generator=g)
    g_n = torch.randn(P, 3, device=dev, generator=g)
    g_sdf = torch.randn(P, 1, device=dev, generator=g)
    sdf_l = [getattr(runner.sdf_network, 'lin%d' % l) for l in range(runner.sdf_network.num_layers - 1)]
    col_l = [getattr(runner.color_network, 'lin%d' % l) for l in range(runner.color_network.num_layers - 1)]
    with torch.no_grad():
        W, b = [m.effective_weight().float() for m in sdf_l], [m.bias.float() for m in sdf_l]
        Wc, bc = [m.effective_weight().float() for m in col_l], [m.bias.float() for m in col_l]
        wbuf, descs, flat = eng.pack(W, b, Wc, bc, want_flat=True)
        Ta, Tb = eng.alloc_tensors(P, dev), eng.alloc_tensors(P, dev)
        for T in (Ta, Tb):
            T['X'].copy_(x); T['DIRS'].copy_(d)
            eng.run_fused_forward(flat, T, P)
        outs = ['DC%d' % l for l in range(eng.nC + 1)] + ['GOUTF', 'ED'] + ['UD%d' % (l + 1) for l in range(eng.nL)] + ['AB%d' % l for l in range(eng.nL)]
        for n in outs:
            Tb[n].fill_(float('nan'))
        rgb = Ta['RGB']
        Ta['DOUT'].copy_(g_rgb * rgb * (1.0 - rgb))
        eng.run('prog_cbwd', descs, wbuf, Ta, P)
        Ta['V'].copy_(g_n + Ta['GNCOL'])
        Ta['GS'].copy_(g_sdf)
        eng.run('prog_sbwd', descs, wbuf, Ta, P)
        eng.run_fused_backward(flat, Tb, P, g_rgb, g_n, g_sdf)
        torch.cuda.synchronize()
        width = {'GOUTF': eng.F, 'ED': eng.E, 'DC%d' % eng.nC: 3}
        for l in range(eng.nL):
            width['UD%d' % (l + 1)] = width['AB%d' % l] = eng.out[l]
        for l in range(eng.nC):
            width['DC%d' % l] = eng.cout[l]
        worst = 0
        for n in outs:
            a, c = Ta[n], Tb[n]
            nt = (P + 31) // 32
            full_a, full_c = a[:nt], c[:nt]
            nanfull = int(torch.isnan(full_c).sum())
            a = a.permute(0, 3, 1, 2).reshape(a.shape[0] * 32, -1)[:P, :width[n]]
            c = c.permute(0, 3, 1, 2).reshape(c.shape[0] * 32, -1)[:P, :width[n]]
            err = float((a - c).abs().max()); sc = float(a.abs().max())
            worst = max(worst, err / max(sc, 1e-12))
            pad = float((full_a - full_c).abs().max())
            print(f'{n:6s} max|prog| {sc:.4e}  max diff {err:.3e}  rel {err/max(sc,1e-12):.2e}  nan(all of the tiles) {nanfull}  diff incl. padding {pad:.3e}')
        print('P', P, 'worst rel', worst)
        Tc = eng.alloc_tensors(P, dev)
        Tc['X'].copy_(x); Tc['DIRS'].copy_(d)
        eng.run_fused_forward(flat, Tc, P)
        for n in outs:
            Tc[n].fill_(float('nan'))
        eng.run_fused_backward_x3(flat, Tc, P, g_rgb, g_n, g_sdf)
        torch.cuda.synchronize()
        worst = 0
        for n in outs:
            a, c = Ta[n], Tc[n]
            nt = (P + 31) // 32
            nanfull = int(torch.isnan(c[:nt]).sum())
            a = a.permute(0, 3, 1, 2).reshape(a.shape[0] * 32, -1)[:P, :width[n]]
            c = c.permute(0, 3, 1, 2).reshape(c.shape[0] * 32, -1)[:P, :width[n]]
            err = float((a - c).abs().max()); sc = float(a.abs().max())
            worst = max(worst, err / max(sc, 1e-12))
            print(f'x3 {n:6s} max|prog| {sc:.4e}  max diff {err:.3e}  rel {err/max(sc,1e-12):.2e}  nan {nanfull}')
        print('x3 P', P, 'worst rel', worst)
for mode in ('prog', 'fused', 'x3'):
    os.environ['VQN_TRAIN_BWD'] = mode
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize(); print(f'VQN_TRAIN_BWD={mode}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step (eager)')
    _C.KernelClock.reset(True)
    step(); torch.cuda.synchronize()
    summ = _C.KernelClock.summary()
    _C.KernelClock.reset(False)
    for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1])[:7]:
        print('   ', k, v)
