"""Captured 2048-point reflectance step against the partial-block size of its contractions (VQN_WGRAD_TPB) and the side stream; and
run-to-run determinism of the captured step (two trainers from the same state must end bit-identical)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'det':
    sys.path.insert(0, ROOT)
    import torch, bench
    dev = torch.device('cuda:0')
    outs = []
    for rep in range(2):
        torch.manual_seed(0)
        model, tr, step = bench.decomp_train_setup(dev, 0, 1, graph=True)
        for _ in range(12): step()
        torch.cuda.synchronize()
        outs.append([p.detach().clone() for p in [model._codebook] + list(model.trainable_variables)])
    same = all(torch.equal(a, b) for a, b in zip(*outs))
    worst = max(float((a - b).abs().max()) for a, b in zip(*outs))
    print('two captured runs from the same state bit-identical:', same, 'max abs diff', worst)
    sys.exit(0)
for env in ({'VQN_WGRAD_TPB': '1'}, {'VQN_WGRAD_TPB': '2'}, {'VQN_WGRAD_TPB': '4'}, {'VQN_WGRAD_TPB': '8'}, {'VQN_WGRAD_TPB': '16'},
            {'VQN_WGRAD_TPB': '4', 'VQN_SIDE_WGRAD': '1'}):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'debug', 'refl_graph_time.py')], env={**os.environ, **env}, capture_output=True, text=True, cwd=ROOT)
    print(env, (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
for env in ({}, {'VQN_SIDE_WGRAD': '1'}):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), 'det'], env={**os.environ, **env}, capture_output=True, text=True, cwd=ROOT)
    print(env, (r.stdout.strip().splitlines() or [r.stderr[-600:]])[-1], flush=True)
