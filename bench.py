#!/usr/bin/env python3
"""bench.py -- rays/sec of the NeuS ray-march hot path on MI355X (driver contract: one JSON line on rank 0).

Workload (BASELINE.json configs[1]): NeuSRenderer.render on synthetic rays of an 800x800 image, full
nets (SDF 8x256 -> 257, colour 4x256), 64 coarse + 64 importance samples per ray (4 up-sampling steps),
fp32 end to end.  A "step" = one render() call over a batch of `--rays` rays already resident in HBM
(default 80,000 = 100 rows of the image; at 8 GPUs one step = one full image, weak scaling).
Ranks shard rays with no data-path collective (gen_geo.py --num_p/--p_i is the reference's own scheme).

Extra objects in the JSON line:
  roofline     -- dominant kernel (vqn_neus_fine_points): algorithmic FLOPs per launch / its average
                  launch duration measured with HIP events on the launch stream, vs the dense f32 MFMA peak
  cpu_baseline -- the faithful torch-CPU oracle (oracle/geo.py, a port of the reference op sequence) timed
                  on this host's cores over a bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

F32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA (= f32 vector peak)

FULL = dict(
    sdf=dict(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5, scale=1.0,
             geometric_init=True, weight_norm=True),
    color=dict(d_feature=256, mode='idr', d_in=9, d_out=3, d_hidden=256, n_layers=4, weight_norm=True,
               multires_view=4, squeeze_out=True),
    renderer=dict(n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4, perturb=1.0),
)


def image_rays(rows, H=800, W=800, fov=0.6911, cam_z=4.0):
    """Pin-hole camera at (0,0,cam_z) looking down -z; returns o, d [len(rows)*W, 3]."""
    f = 0.5 * W / math.tan(0.5 * fov)
    j, i = np.meshgrid(np.asarray(rows, np.float64), np.arange(W, dtype=np.float64), indexing='ij')
    d = np.stack([(i - 0.5 * W + 0.5) / f, -(j - 0.5 * H + 0.5) / f, -np.ones_like(i)], -1).reshape(-1, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile(np.array([[0.0, 0.0, cam_z]]), (d.shape[0], 1))
    return o.astype(np.float32), d.astype(np.float32)


def macs_per_point(sdf, col):
    """MACs of one SDF forward and one colour forward, from the layer shapes."""
    m_sdf = 0
    for l in range(sdf.num_layers - 1):
        lin = getattr(sdf, f'lin{l}')
        m_sdf += lin.bias.numel() * (lin.weight_v.shape[1] if lin.weight_norm else lin.weight.shape[1])
    m_col = 0
    for l in range(col.num_layers - 1):
        lin = getattr(col, f'lin{l}')
        m_col += lin.bias.numel() * (lin.weight_v.shape[1] if lin.weight_norm else lin.weight.shape[1])
    return m_sdf, m_col


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--rays', type=int, default=80000, help='rays per step per GPU')
    ap.add_argument('--cpu-rays', type=int, default=384, help='rays of the bounded CPU-baseline sample')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run'
    assert torch.cuda.is_available(), 'bench.py needs an MI355X (no CPU fallback)'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('nccl', device_id=dev)

    from vqnerf_release_amd import _C
    from vqnerf_release_amd.geo.models.fields import SDFNetwork, RenderingNetwork, SingleVarianceNetwork
    from vqnerf_release_amd.geo.models.renderer import NeuSRenderer
    _C.lib()

    # ---- random-init weights of the shipped architecture (same seed on every rank) ----
    torch.manual_seed(0)
    sdf = SDFNetwork(**FULL['sdf'])
    col = RenderingNetwork(**FULL['color'])
    var = SingleVarianceNetwork(0.3)
    state = {'sdf': {k: v.clone() for k, v in sdf.state_dict().items()},
             'col': {k: v.clone() for k, v in col.state_dict().items()}}
    sdf, col, var = sdf.to(dev), col.to(dev), var.to(dev)
    ren = NeuSRenderer(None, sdf, var, col, **FULL['renderer'])
    S_f = FULL['renderer']['n_samples'] + FULL['renderer']['n_importance']

    # ---- this rank's rays: consecutive rows of the 800x800 image, resident in HBM ----
    n_rows = (args.rays + 799) // 800
    rows = (np.arange(n_rows) + rank * n_rows) % 800
    o_np, d_np = image_rays(rows)
    o_np, d_np = o_np[:args.rays], d_np[:args.rays]
    o, d = torch.tensor(o_np, device=dev), torch.tensor(d_np, device=dev)
    near = torch.full((args.rays, 1), 2.0, device=dev)
    far = torch.full((args.rays, 1), 6.0, device=dev)
    bg = torch.ones(1, 3, device=dev)

    def step():
        with torch.no_grad():
            return ren.render(o, d, near, far, 2.0, perturb_overwrite=0, background_rgb=bg, cos_anneal_ratio=1.0)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    barrier()
    _C.KernelClock.reset(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    clock = _C.KernelClock.summary()
    _C.KernelClock.reset(False)
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(out['color_fine']).all()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    total_rays = args.rays * world * args.steps
    value = total_rays / dt
    m_sdf, m_col = macs_per_point(sdf, col)
    # dominant kernel: one fine launch evaluates, per point, one SDF forward, one reverse (input-gradient)
    # sweep of the same MACs, and one colour forward
    n_fine, ms_fine = clock['vqn_neus_fine_points']
    flop_fine = 2.0 * (2 * m_sdf + m_col) * args.rays * S_f
    avg_ms = ms_fine / n_fine
    achieved = flop_fine / (avg_ms * 1e-3) / 1e12
    result = {
        'metric': 'rays/sec (render) 800x800, 64+64 samples/ray, NeuS SDF 8x256 + colour 4x256',
        'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'nerf/hotdog-shaped NeuS render: 800x800 pin-hole rays, n_samples=64, n_importance=64, '
                               'up_sample_steps=4, sdf 8x256 (skip 4, posenc 6), colour 4x256 (idr, posenc_view 4), '
                               'random-init weights', 'rays_per_step_per_gpu': args.rays, 'parallelism': f'rays x{world}'},
        'roofline': {'bound': 'mfma', 'kernel': 'neus_points_kernel<FINE> (vqn_neus_fine_points)', 'achieved': achieved,
                     'peak': F32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / F32_MFMA_PEAK_TFLOPS,
                     'traffic': None, 'avg_launch_ms': avg_ms, 'flop_per_launch': flop_fine,
                     'macs_per_point': {'sdf': m_sdf, 'colour': m_col}},
        'kernel_ms_per_step': {k: v[1] / args.steps for k, v in sorted(clock.items())},
    }

    if world == 1 and not args.no_cpu_baseline:
        from oracle import geo as og                      # CPU-baseline leg only (test infrastructure)
        cfg = dict(og.FULL_CFG)
        p_sdf = {k: v.float() for k, v in state['sdf'].items()}
        p_col = {k: v.float() for k, v in state['col'].items()}
        n_cpu = args.cpu_rays
        sel = np.linspace(0, args.rays - 1, n_cpu).astype(np.int64)      # spread over the step's rays
        oc, dc = torch.tensor(o_np[sel]), torch.tensor(d_np[sel])
        nc, fc = torch.full((n_cpu, 1), 2.0), torch.full((n_cpu, 1), 6.0)
        cores = os.cpu_count() or 1
        try:
            cores = len(os.sched_getaffinity(0))
        except Exception:
            pass
        # a 1-GPU box shares its host: the CPU share of one GPU is 16 cores (more threads only oversubscribe)
        cores = int(os.environ.get('VQN_CPU_THREADS', min(cores, 16)))
        torch.set_num_threads(cores)
        og.render(p_sdf, p_col, torch.tensor(0.3), cfg, oc[:32], dc[:32], nc[:32], fc[:32], 2.0,
                  background_rgb=torch.ones(1, 3), cos_anneal_ratio=1.0)          # warm-up
        t0 = time.perf_counter()
        ref = og.render(p_sdf, p_col, torch.tensor(0.3), cfg, oc, dc, nc, fc, 2.0, background_rgb=torch.ones(1, 3),
                        cos_anneal_ratio=1.0)
        cpu_dt = time.perf_counter() - t0
        got = out['color_fine'][torch.tensor(sel, device=dev)].cpu()
        mse = float(((got - ref['color_fine'].detach()) ** 2).mean())
        result['cpu_baseline'] = {'value': n_cpu / cpu_dt, 'unit': 'rays/s', 'cores': cores, 'kind': 'port',
                                  'sample': f'{n_cpu} rays of the same step (same weights), one oracle.geo.render call, '
                                            f'{cpu_dt:.1f} s, torch {torch.__version__} CPU fp32'}
        result['psnr_vs_oracle_db'] = -10.0 * math.log10(mse + 1e-20)
        result['speedup_vs_cpu'] = value / (n_cpu / cpu_dt)
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
