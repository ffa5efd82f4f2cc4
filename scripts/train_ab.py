"""Training A/B of the exact-split (x3) kernels against the f32-input MFMA kernels, sized to decide something (VERDICT r04 next #5 iv).

Round 4's A/B (2,000 steps, one seed per arm, 15 dB) could not tell the arms apart from trajectory noise.  This one: the same geo trainer
(geo/nerf_runner.Runner(graph=True): full-size networks, 2560 rays per step, the synthetic 8-view set of bench.py, views 0..6 trained,
view 7 held out) for >= 20,000 steps from >= 3 seeds per arm --
    x3 : vqn_neus_train_fwd_x3 / _bwd_x3, x3 up-sampling passes, bf16x3 contraction (the default)
    f32: VQN_TRAIN_FWD=fused VQN_TRAIN_BWD=fused VQN_TRAIN_COARSE=f32 VQN_WGRAD=f32
-- held-out PSNR (200 x 200 pixels of view 7, rendered on the f32 kernels in both arms), training-view PSNR, s_val, loss curve.
Gate for x3 staying the training default: the arms' mean held-out PSNR within ONE pooled standard deviation of each other (and no arm
with a diverged run); otherwise f32 training becomes the default.

    python scripts/train_ab.py run  <steps> <arm>:<seed> [<arm>:<seed> ...]   -> gpurun_out/train_ab/<arm>_s<seed>.json (one child per run)
    python scripts/train_ab.py report                                         -> profiles/r05_train_ab.json from whatever runs are there
A gpurun call lasts at most 20 minutes: two runs per call (`run 20000 x3:0 f32:0`), the report is built on the CPU afterwards."""
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ARMS = {'x3': {}, 'f32': {'VQN_TRAIN_FWD': 'fused', 'VQN_TRAIN_BWD': 'fused', 'VQN_TRAIN_COARSE': 'f32', 'VQN_WGRAD': 'f32'}}
OUT = os.path.join(ROOT, 'gpurun_out', 'train_ab')


def child(arm, seed, steps):
    import torch
    import bench
    from vqnerf_release_amd.geo.nerf_runner import Runner, SyntheticDataset
    dev = torch.device('cuda:0')
    torch.manual_seed(seed)                                      # initial weights; the pixel draws follow the dataset's generator below
    ds = SyntheticDataset(device=dev, n_images=8, seed=0)        # the SAME scene for every run
    if hasattr(ds, 'gen'):
        ds.gen.manual_seed(1000 + seed)
    runner = Runner(conf_text=bench.full_conf_text(2560), case=f'ab_{arm}_{seed}', dataset=ds, device=dev, graph=True)
    runner.update_learning_rate()

    def view_psnr(idx, n=200):
        ys, xs = torch.meshgrid(torch.linspace(0, ds.H - 1, n, device=dev), torch.linspace(0, ds.W - 1, n, device=dev), indexing='ij')
        px, py = xs.reshape(-1).round(), ys.reshape(-1).round()
        o, fwd, right, up = ds._frame(idx)
        d = fwd[None] + ((px - 0.5 * ds.W + 0.5) / ds.focal)[:, None] * right[None] - ((py - 0.5 * ds.H + 0.5) / ds.focal)[:, None] * up[None]
        d = d / d.norm(dim=-1, keepdim=True)
        b = (o[None] * d).sum(-1)
        hit = (b * b - (o.dot(o) - 0.36)) > 0
        rgb = torch.where(hit[:, None], 0.5 + 0.5 * torch.sin(torch.stack([px, py, px + py], -1) * 0.01), torch.ones(px.numel(), 3, device=dev))
        oo = o[None].expand(px.numel(), 3).contiguous()
        near, far = ds.near_far_from_sphere(oo, d)
        runner.renderer.matrix_mode = 'f32'                      # (the validation render on the f32 kernels in both arms)
        with torch.no_grad():
            r = runner.renderer.render(oo, d.contiguous(), near, far, ds.max_radius, perturb_overwrite=0, background_rgb=torch.ones(1, 3, device=dev),
                                       cos_anneal_ratio=1.0)
        mse = float(((r['color_fine'] - rgb) ** 2).mean())
        return -10.0 * math.log10(mse + 1e-20)
    hist = {'step': [], 'loss': [], 'color_loss': [], 'eikonal_loss': [], 'variance': [], 'psnr_held_out_db': {}}
    t0 = time.time()
    for it in range(steps):
        st = runner.train_step(ds.gen_random_rays_at(it % 7, 2560))
        if it % 500 == 499 or it == 0:
            hist['step'].append(it + 1)
            for k in ('loss', 'color_loss', 'eikonal_loss'):
                hist[k].append(float(st[k]))
            hist['variance'].append(float(runner.deviation_network.variance))
        if (it + 1) % 5000 == 0:
            hist['psnr_held_out_db'][str(it + 1)] = view_psnr(7)
            print(f'[{arm} seed {seed}] step {it + 1}: loss {float(st["loss"]):.4f}, held-out PSNR {hist["psnr_held_out_db"][str(it + 1)]:.2f} dB, '
                  f'{(time.time() - t0) / (it + 1) * 1e3:.1f} ms/step', flush=True)
    torch.cuda.synchronize()
    wall = time.time() - t0
    res = {'arm': arm, 'seed': seed, 'env': ARMS[arm], 'steps': steps, 'wall_s': wall, 'ms_per_step': wall / steps * 1e3, 'history': hist,
           'psnr_held_out_view7_db': view_psnr(7), 'psnr_training_view0_db': view_psnr(0), 'captured': runner._cap is not None,
           's_val_final': float(torch.exp(runner.deviation_network.variance * 10.0)), 'last_train_backend': runner.renderer.last_train_backend,
           'finite': bool(all(torch.isfinite(p).all() for p in runner.sdf_network.parameters()))}
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, f'{arm}_s{seed}.json'), 'w') as f:
        json.dump(res, f)


def report():
    import numpy as np
    runs = {a: [] for a in ARMS}
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith('.json'):
            with open(os.path.join(OUT, fn)) as f:
                r = json.load(f)
            runs[r['arm']].append(r)
    stat = {}
    for a, rs in runs.items():
        v = np.array([r['psnr_held_out_view7_db'] for r in rs])
        t = np.array([r['psnr_training_view0_db'] for r in rs])
        stat[a] = {'n': len(rs), 'seeds': [r['seed'] for r in rs], 'steps': sorted({r['steps'] for r in rs}),
                   'psnr_held_out_db': [float(x) for x in v], 'psnr_held_out_mean_db': float(v.mean()) if len(v) else None,
                   'psnr_held_out_std_db': float(v.std(ddof=1)) if len(v) > 1 else None,
                   'psnr_training_view_mean_db': float(t.mean()) if len(t) else None,
                   'psnr_training_view_std_db': float(t.std(ddof=1)) if len(t) > 1 else None,
                   's_val_final': [r['s_val_final'] for r in rs], 'ms_per_step': [r['ms_per_step'] for r in rs],
                   'all_finite': all(r['finite'] for r in rs), 'all_captured': all(r['captured'] for r in rs),
                   'psnr_held_out_curve_db': [r['history']['psnr_held_out_db'] for r in rs]}
    rep = {'what': 'geo training A/B, exact-split (x3) kernels vs f32-input MFMA kernels: full-size NeuS networks, 2560 rays / step, synthetic 8-view '
                   'set (views 0..6 trained, view 7 held out at 200 x 200 pixels), Runner(graph=True) in both arms, one scene, seeds differ in '
                   'initial weights and pixel draws',
           'arms': stat}
    a, b = stat['x3'], stat['f32']
    if a['n'] > 1 and b['n'] > 1:
        pooled = math.sqrt(((a['n'] - 1) * a['psnr_held_out_std_db'] ** 2 + (b['n'] - 1) * b['psnr_held_out_std_db'] ** 2) / (a['n'] + b['n'] - 2))
        gap = a['psnr_held_out_mean_db'] - b['psnr_held_out_mean_db']
        rep['gate'] = {'mean_gap_x3_minus_f32_db': gap, 'pooled_std_db': pooled, 'means_within_one_pooled_std': abs(gap) <= pooled,
                       'both_arms_finite': a['all_finite'] and b['all_finite'],
                       'x3_stays_training_default': bool(abs(gap) <= pooled and a['all_finite'])}
    with open(os.path.join(ROOT, 'profiles', 'r05_train_ab.json'), 'w') as f:
        json.dump(rep, f, indent=1)
    print(json.dumps({k: v for k, v in rep.items()}, indent=1)[:3000])


if __name__ == '__main__':
    if sys.argv[1] == 'child':
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    elif sys.argv[1] == 'run':
        steps = int(sys.argv[2])
        for spec in sys.argv[3:]:
            arm, seed = spec.split(':')
            e = dict(os.environ)
            e.update(ARMS[arm])
            t0 = time.time()
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child', arm, seed, str(steps)], env=e, check=True)
            print(f'{spec} done in {time.time() - t0:.0f} s', flush=True)
    elif sys.argv[1] == 'report':
        report()
