"""CPU: tests/golden/kmeans.npz (outputs of the reference's torch_kmeans.py, see oracle/gen_golden_kmeans.py) is a fixed
point of Lloyd's iteration on the inputs the tests re-make, and the mirror's `initialize` draws the reference's rows."""
import os

import numpy as np
import pytest
import torch

from oracle.gen_golden_kmeans import CASES, kmeans_inputs


@pytest.mark.parametrize('name', list(CASES))
def test_fixture_is_a_lloyd_fixed_point(golden_dir, name):
    from vqnerf_release_amd.decomp.nerfactor.util.torch_kmeans import initialize, pairwise_distance, pairwise_cosine
    g = np.load(os.path.join(golden_dir, 'kmeans.npz'))
    n, K, D, noise, distance, seed = CASES[name]
    X = kmeans_inputs(name)
    assert int(g[f'{name}_seed']) == seed and X.shape == (n, D)
    np.testing.assert_array_equal(initialize(torch.tensor(X), K, seed).numpy(), g[f'{name}_init'])
    ids, c = g[f'{name}_ids'], g[f'{name}_centres']
    assert ids.shape == (n,) and c.shape == (K, D) and len(np.unique(ids)) == K
    fn = pairwise_distance if distance == 'euclidean' else pairwise_cosine
    # the ids the reference returned were taken against the centres of the step before its last update, whose shift was
    # below tol = 1e-4: re-assigning against the returned centres changes (almost) nothing
    again = torch.argmin(fn(torch.tensor(X), torch.tensor(c)).reshape(n, K), 1).numpy()
    assert (again != ids).mean() <= 1e-3
    means = np.stack([X[ids == k].mean(0) for k in range(K)])
    np.testing.assert_allclose(means, c, rtol=0, atol=1e-5)
