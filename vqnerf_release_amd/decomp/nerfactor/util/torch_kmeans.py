"""Lloyd k-means for the codebook initialisation: mirror of decomp/nerfvq_nfr3/nerfactor/util/torch_kmeans.py:7-163
(`initialize`, `kmeans`, `kmeans_predict`, `pairwise_distance`, `pairwise_cosine`; same arguments and return values).

The reference runs this on the CPU over ~10^5 latent vectors (train_nfr.py:471-488).  Here the assignment step is the VQ
nearest-code kernel (`vqn_vq_assign`, same squared-L2 argmin with lowest-index tie-break) and the update step is the EMA
statistics kernel (`vqn_vq_ema_stats`: per-cluster sums and counts), both on the device; only the scalar convergence test
comes back to the host each iteration.  Differences: an empty cluster keeps its previous centre (the reference's
`mean` of an empty selection is NaN); distances use |x|^2 - 2 x.c + |c|^2 rather than sum((x - c)^2)."""
import numpy as np
import torch

from vqnerf_release_amd import _C


def initialize(X, num_clusters, seed):
    """same draw as the reference (torch_kmeans.py:7-19): numpy RNG, `choice` without replacement"""
    np.random.seed(seed)
    indices = np.random.choice(len(X), num_clusters, replace=False)
    return X[torch.as_tensor(indices, device=X.device)]


def pairwise_distance(data1, data2, device=None):
    a, b = data1.unsqueeze(1), data2.unsqueeze(0)
    return ((a - b) ** 2.0).sum(-1).squeeze()


def pairwise_cosine(data1, data2, device=None):
    a, b = data1.unsqueeze(1), data2.unsqueeze(0)
    a, b = a / a.norm(dim=-1, keepdim=True), b / b.norm(dim=-1, keepdim=True)
    return 1 - (a * b).sum(-1).squeeze()


def _kernel_ok(X, k):
    return X.is_cuda and X.shape[1] % 4 == 0 and k <= 128


def _assign(X, centers, distance):
    if distance == 'euclidean' and _kernel_ok(X, centers.shape[0]):
        idx, _, _ = _C.vq_assign(X, centers.t().contiguous(), want_quant=False)
        return idx
    fn = pairwise_distance if distance == 'euclidean' else pairwise_cosine
    out = []
    for s in range(0, X.shape[0], 65536):                      # the [n, k, d] broadcast of the reference, in slabs
        out.append(torch.argmin(fn(X[s:s + 65536], centers).reshape(-1, centers.shape[0]), dim=1))
    return torch.cat(out)


def kmeans(X, num_clusters, distance='euclidean', tol=1e-4, device=None, seed=1, max_iter=10000):
    """-> (cluster ids [n] int64, cluster centres [num_clusters, d]) on X's device (the reference returns CPU tensors)."""
    if distance not in ('euclidean', 'cosine'):
        raise NotImplementedError
    X = X.float()
    if device is not None:
        X = X.to(device)
    _C.require_device(X, 'kmeans')
    X = X.contiguous()
    state = initialize(X, num_clusters, seed).clone()
    for _ in range(max_iter):
        choice = _assign(X, state, distance)
        prev = state.clone()
        if _kernel_ok(X, num_clusters):
            counts, sums = _C.vq_ema_stats(X, choice, num_clusters)           # sums [d, k]
            mean = sums.t() / counts.clamp(min=1.0)[:, None]
            state = torch.where((counts > 0)[:, None], mean, prev)
        else:
            onehot = torch.nn.functional.one_hot(choice, num_clusters).float()
            counts = onehot.sum(0)
            state = torch.where((counts > 0)[:, None], (onehot.t() @ X) / counts.clamp(min=1.0)[:, None], prev)
        center_shift = torch.sqrt(((state - prev) ** 2).sum(1)).sum()
        if float(center_shift) ** 2 < tol:
            break
    return choice, state


def kmeans_predict(X, cluster_centers, distance='euclidean', device=None):
    if distance not in ('euclidean', 'cosine'):
        raise NotImplementedError
    X = X.float()
    if device is not None:
        X = X.to(device)
    _C.require_device(X, 'kmeans_predict')
    return _assign(X.contiguous(), cluster_centers.to(X.device).float().contiguous(), distance)
