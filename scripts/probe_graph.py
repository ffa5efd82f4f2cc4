"""Dev probe: which pieces of the reflectance-model training step survive HIP-graph capture + replay.
`python scripts/probe_graph.py` runs every piece in its own child process (a crash in one does not hide the others);
`python scripts/probe_graph.py <piece>` runs one."""
import faulthandler
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PIECES = ['trainer', 'trainer2', 'testfn', 'torch', 'vq_assign', 'vq_ema', 'enc_fwd', 'enc', 'heads', 'shade_fwd', 'shade', 'quantise', 'adam', 'full']


def child(piece):
    faulthandler.enable()
    import numpy as np, torch
    if piece == 'testfn':
        from tests import test_gpu_train
        test_gpu_train.test_decomp_trainer_graph_replays_the_eager_step()
        print('testfn ok')
        return
    import bench
    from vqnerf_release_amd import _C
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(1)
    cfg = config_from_dict(bench.DECOMP_INI)
    model = get_model_class('vq_nfr')(cfg)
    model.build_nets(device=dev, seed=0).to(dev)
    cb = rng.uniform(0, 1, (15, 256)).astype(np.float32)
    model.set_codebook(cb / np.linalg.norm(cb, axis=1, keepdims=True))
    model.set_light(rng.uniform(0, 1, (16, 32, 3)).astype(np.float32))
    model.get_codebook(); _ = model.light
    n = int(os.environ.get('PROBE_N', '2048'))
    xyz = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1)
    nrm = torch.nn.functional.normalize(xyz + 0.1 * torch.randn(n, 3, device=dev), dim=-1)
    rayo = torch.tensor([[0, 0, 4.0]], device=dev).repeat(n, 1)
    one = torch.ones(n, 1, device=dev)
    lvis = (torch.rand(n, 512, device=dev) < 0.7).float()
    batch = (['v'], torch.zeros(n, 2, device=dev), rayo, torch.zeros(n, 3, device=dev), torch.rand(n, 3, device=dev), one, one.clone(),
             xyz, nrm, lvis)
    z = torch.nn.functional.normalize(torch.rand(n, 256, device=dev), dim=-1).requires_grad_(True)
    C = model.get_codebook().detach()
    opt, _, clip = train_nfr.make_optimizer(cfg, model.trainable_variables, capturable=True)
    tr = train_nfr.Trainer(model, opt, clip=clip)
    model.assume_foreground = True
    mats = lambda: [(torch.rand(n, 3, device=dev).requires_grad_(True), torch.rand(n, 3, device=dev).requires_grad_(True),
                     torch.rand(n, 1, device=dev).requires_grad_(True)) for _ in range(2)]
    M = mats()

    def zero():
        for p in model.trainable_variables:
            p.grad = None

    def f_torch():
        (xyz * 2 + nrm).sum()

    def f_vq_assign():
        _C.vq_assign(z.detach(), C)

    def f_vq_ema():
        idx = _C.vq_assign(z.detach(), C, want_quant=False)[0]
        _C.vq_ema_stats(z.detach(), idx, 15)

    def f_enc_fwd():
        with torch.no_grad():
            model._pred_enc_at(xyz)

    def f_enc():
        model._pred_enc_at(xyz).sum().backward()

    def f_heads():
        sum(o.sum() for o in model._all_heads(z, 'main')).backward()

    def f_shade_fwd():
        with torch.no_grad():
            model._shade_or_render(xyz, nrm, rayo, lvis, [tuple(t.detach() for t in m) for m in M], split=True)

    def f_shade():
        sh = model._shade_train(xyz, nrm, rayo, lvis, M)
        (sh['rgb'][0].sum() + sh['rgb'][1].sum()).backward()

    def f_quantise():
        model._quantise(z.detach(), 'train', None)

    def f_adam():
        opt.step()

    def f_full():
        tr._step(batch, 1024, None, None)

    if piece in ('trainer', 'trainer2'):
        if piece == 'trainer2':
            for _ in range(3):
                tr.train_iter(batch, 1024)
            model2 = get_model_class('vq_nfr')(cfg)
            model2.build_nets(device=dev, seed=0).to(dev)
            model2.set_codebook(cb / np.linalg.norm(cb, axis=1, keepdims=True))
            model2.set_light(rng.uniform(0, 1, (16, 32, 3)).astype(np.float32))
            model2.get_codebook(); _ = model2.light
            model = model2
            opt, _, clip = train_nfr.make_optimizer(cfg, model.trainable_variables, capturable=True)
        tg = train_nfr.Trainer(model, opt, clip=clip, graph=True)
        for i in range(5):
            wl, _, _ = tg.train_iter(batch, 1024)
            print(piece, i, float(wl), flush=True)
        print(piece, 'ok', flush=True)
        return
    fn = locals()['f_' + piece]
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    if piece not in ('full',):
        zero()
        if piece == 'adam':
            for p in model.trainable_variables:
                p.grad = torch.zeros_like(p)
    print(piece, 'eager ok', flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    print(piece, 'captured', flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print(piece, f'replay ok {e0.elapsed_time(e1) / 10:.3f} ms', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] in PIECES:
        child(sys.argv[1])
    else:
        for p in PIECES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), p], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                               timeout=300)
            tail = '\n'.join(r.stdout.strip().splitlines()[-6:])
            print(f'=== {p}: rc={r.returncode}\n{tail}', flush=True)
