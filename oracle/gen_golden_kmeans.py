#!/usr/bin/env python
"""Generate tests/golden/kmeans.npz by RUNNING THE REAL REFERENCE k-means
(/root/reference/decomp/nerfvq_nfr3/nerfactor/util/torch_kmeans.py: needs only numpy / torch / tqdm, so it imports here).

Build-container only.  Inputs come from numpy seeds (oracle.kmeans_inputs below, re-made by the tests); only the reference's
OUTPUTS are stored: cluster ids, centres, `kmeans_predict` ids, and the number of Lloyd iterations it took.

    python oracle/gen_golden_kmeans.py        # writes tests/golden/kmeans.npz
"""
import contextlib
import importlib.util
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('VQNERF_REFERENCE', '/root/reference')
GOLD = os.path.join(ROOT, 'tests', 'golden')

# Seeds are chosen so that the REFERENCE terminates: when a cluster runs empty its `mean` is NaN, every later centre shift is
# NaN and the reference's `while True` never exits (torch_kmeans.py:66-90) -- e.g. seed 1 on 'k8'.  `_terminates` checks
# that on a bounded replica of the loop before the real one is started.  The seeds below also draw one initial centre from
# every true cluster, so the iteration settles in 2-3 steps on well-separated clusters: a seed that splits one cluster between
# two centres converges too, but to a partition that an fp32 rounding difference in an early step reshuffles (chaotic, no
# fixture value worth pinning).
CASES = {                       # name: (n, K, D, noise, distance, seed)
    'k8': (4000, 8, 256, 0.05, 'euclidean', 6),
    'k15': (3000, 15, 256, 0.08, 'euclidean', 2),        # the shipped codebook size (vq_nfr.ini:113), latent-like inputs
    'k64': (6000, 64, 256, 0.04, 'euclidean', 35),       # BASELINE configs[2]
    'k8cos': (2000, 8, 64, 0.05, 'cosine', 5),
}


def kmeans_inputs(name):
    """Seeded clustered data in (0, 1)^D, l2-normalised rows for the latent-like cases (what train_nfr.z_cluster feeds)."""
    n, K, D, noise, distance, seed = CASES[name]
    rng = np.random.default_rng(100 + len(name) + K)
    centres = rng.uniform(0, 1, (K, D))
    X = centres[rng.integers(0, K, n)] + noise * rng.normal(size=(n, D))
    if name in ('k15', 'k64'):
        X = np.abs(X)
        X = X / np.linalg.norm(X, axis=1, keepdims=True)
    return X.astype(np.float32)


def _terminates(ref, X, K, distance, seed, max_iter=200):
    fn = ref.pairwise_distance if distance == 'euclidean' else ref.pairwise_cosine
    with contextlib.redirect_stdout(io.StringIO()):
        st = ref.initialize(X, K, seed)
    for _ in range(max_iter):
        ch = torch.argmin(fn(X, st), 1)
        if (torch.bincount(ch, minlength=K) == 0).any():
            return False
        pre = st.clone()
        for i in range(K):
            st[i] = X[ch == i].mean(0)
        if float(torch.sum(torch.sqrt(torch.sum((st - pre) ** 2, 1))) ** 2) < 1e-4:
            return True
    return False


def main():
    spec = importlib.util.spec_from_file_location(
        'ref_torch_kmeans', os.path.join(REF, 'decomp', 'nerfvq_nfr3', 'nerfactor', 'util', 'torch_kmeans.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    torch.set_num_threads(8)
    out = {}
    for name, (n, K, D, noise, distance, seed) in CASES.items():
        X = torch.tensor(kmeans_inputs(name))
        assert _terminates(ref, X, K, distance, seed), f'{name}: the reference would loop forever on this seed (empty cluster)'
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            ids, centres = ref.kmeans(X, K, distance=distance, seed=seed)
            pred = ref.kmeans_predict(X[:500], centres, distance=distance)
            init = ref.initialize(X, K, seed)
        out[f'{name}_ids'] = ids.numpy()
        out[f'{name}_centres'] = centres.numpy()
        out[f'{name}_predict'] = pred.numpy()
        out[f'{name}_init'] = init.numpy()
        out[f'{name}_seed'] = np.asarray(seed)
        assert np.isfinite(out[f'{name}_centres']).all(), name          # no empty cluster (the reference would give NaN there)
        # well-separated: the second-nearest centre is clearly further away, so ids are not rounding-sensitive
        d = torch.cdist(X.double(), centres.double()).numpy()
        d.sort(1)
        out[f'{name}_min_gap'] = np.asarray((d[:, 1] - d[:, 0]).min())
        print(name, 'clusters used:', len(np.unique(out[f'{name}_ids'])), 'min top-2 gap:', float(out[f'{name}_min_gap']))
    np.savez_compressed(os.path.join(GOLD, 'kmeans.npz'), **out)
    print('done ->', os.path.join(GOLD, 'kmeans.npz'))


if __name__ == '__main__':
    main()
