"""GPU: k-means codebook init on the VQ kernels vs a plain numpy Lloyd iteration with the reference's initial draw."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _numpy_kmeans(X, k, seed, tol=1e-4):
    np.random.seed(seed)
    state = X[np.random.choice(len(X), k, replace=False)].copy()
    while True:
        d = ((X[:, None, :] - state[None]) ** 2).sum(-1)
        choice = d.argmin(1)
        prev = state.copy()
        for i in range(k):
            if (choice == i).any():
                state[i] = X[choice == i].mean(0)
        if np.sqrt(((state - prev) ** 2).sum(1)).sum() ** 2 < tol:
            return choice, state


def test_kmeans_matches_lloyd_reference(tmp_path):
    from vqnerf_release_amd.decomp.nerfactor.util.torch_kmeans import kmeans, kmeans_predict, initialize
    from vqnerf_release_amd.decomp.nerfactor import train_nfr
    rng = np.random.default_rng(0)
    K, D = 8, 256
    true_c = rng.uniform(0, 1, (K, D))
    X = (true_c[rng.integers(0, K, 4000)] + 0.05 * rng.normal(size=(4000, D))).astype(np.float32)
    Xt = torch.tensor(X).cuda()
    np.random.seed(1)
    want_idx = np.random.choice(len(X), K, replace=False)
    assert torch.equal(initialize(Xt, K, 1), Xt[torch.tensor(want_idx).cuda()])       # the reference's initial draw
    ids, centers = kmeans(Xt, K, seed=1)
    ref_ids, ref_centers = _numpy_kmeans(X.astype(np.float64), K, 1)
    np.testing.assert_array_equal(ids.cpu().numpy(), ref_ids)
    np.testing.assert_allclose(centers.cpu().numpy(), ref_centers, rtol=0, atol=2e-5)
    assert torch.equal(kmeans_predict(Xt, centers), ids)
    out = train_nfr.z_cluster(None, [X[:2000], X[2000:]], str(tmp_path / 'c.npy'), K, seed=1)
    assert out.shape == (K, D) and np.load(tmp_path / 'c.npy').shape == (K, D)
