"""GPU parity: vqn_vq_assign / vqn_vq_ema_stats / VectorQuantizerEMA against the oracle.

Bar: indices and distances BIT-EXACT against oracle/vq_strict.c (same fixed fmaf order), indices
identical to the fp64 statement wherever the fp64 top-2 gap exceeds 1e-5; EMA sums within 1e-5
relative of the correctly-rounded (double-accumulated) sums."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _data(N, D, K, seed):
    rng = np.random.default_rng(seed)
    z = 1.0 / (1.0 + np.exp(-rng.normal(size=(N, D))))           # sigmoid outputs, like the encoder
    x = (z / np.linalg.norm(z, axis=1, keepdims=True)).astype(np.float32)
    C = rng.uniform(0, 1, (D, K)).astype(np.float32)
    C = (C / np.linalg.norm(C, axis=0, keepdims=True)).astype(np.float32)
    return x, C


@pytest.mark.parametrize('N,D,K', [(1000, 256, 15), (4099, 256, 16), (777, 256, 8), (3000, 256, 64),
                                   (513, 64, 16), (100, 20, 7), (1, 256, 15), (17, 256, 128)])
def test_assign_bit_exact_vs_strict_oracle(N, D, K):
    from oracle import vq_strict as vs
    from vqnerf_release_amd import _C
    x, C = _data(N, D, K, seed=N + K)
    idx, quant, dist = _C.vq_assign(torch.tensor(x).cuda(), torch.tensor(C).cuda(), want_quant=True, want_dist=True)
    ridx, rdist, rquant = vs.assign(x, C)
    np.testing.assert_array_equal(dist.cpu().numpy(), rdist)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(quant.cpu().numpy(), rquant)


def _adversarial(N, D, K, kind, seed):
    """Inputs that stress the candidate filter of vq_assign_split_kernel (K in 17..64, indices only)."""
    rng = np.random.default_rng(seed)
    x, C = _data(N, D, K, seed)
    if kind == 'duplicates':                       # exact ties between code k and k + K/2 -> the lowest index, through the exact passes
        C[:, K // 2:2 * (K // 2)] = C[:, :K // 2]
    elif kind == 'clusters':                       # 6 codes within an ulp or two of each other: more than 4 candidates -> plain f32 path
        for j in range(1, 6):
            C[:, j] = C[:, 0] * np.float32(1.0 + j * 2.0 ** -23)
        x[::3] = C[:, 0][None] * rng.uniform(0.5, 1.5, (len(x[::3]), 1)).astype(np.float32)
    elif kind == 'midpoints':                      # rows half way between two codes (and exactly ON a code)
        a, b = rng.integers(0, K, N), rng.integers(0, K, N)
        x = (0.5 * (C[:, a] + C[:, b])).T.astype(np.float32).copy()
        x[::5] = C[:, a[::5]].T
    elif kind == 'scaled':                         # unnormalised rows and codes over six decades
        x = x * (10.0 ** rng.uniform(-3, 3, (N, 1))).astype(np.float32)
        C = C * (10.0 ** rng.uniform(-1, 1, (1, K))).astype(np.float32)
    elif kind == 'tiny':                           # rows far below the f16 range (an f16 pair has an ABSOLUTE floor): exact passes decide
        x = x * (10.0 ** rng.uniform(-12, -4, (N, 1))).astype(np.float32)
    elif kind == 'huge':                           # elements beyond the f16 range in the rows (some) and in the codebook (all groups)
        x[::4] *= np.float32(3.0e5)
        C = C * np.float32(1.0e5)
    elif kind == 'zeros':
        x[::2] = 0.0
        C[:, 3] = 0.0
    elif kind == 'signed':
        x = rng.normal(size=(N, D)).astype(np.float32)
        C = rng.normal(size=(D, K)).astype(np.float32)
    return np.ascontiguousarray(x, np.float32), np.ascontiguousarray(C, np.float32)


@pytest.mark.parametrize('kind', ['plain', 'duplicates', 'clusters', 'midpoints', 'scaled', 'tiny', 'huge', 'zeros', 'signed'])
@pytest.mark.parametrize('N,D,K', [(3000, 256, 64), (1000, 256, 33), (1000, 256, 17), (2049, 64, 32), (500, 20, 40), (333, 252, 64)])
def test_prefiltered_assign_is_bit_exact_vs_strict_oracle(N, D, K, kind):
    """K in 17..64 without a distance output runs vq_assign_split_kernel (f16-pair prefilter + exact evaluation of the candidates):
    the indices and the gathered rows must be those of the defined arithmetic, including exact ties (lowest index)."""
    from oracle import vq_strict as vs
    from vqnerf_release_amd import _C
    x, C = _adversarial(N, D, K, kind, seed=N + K)
    assert _C.vq_assign_variant(D, K) == 1 and _C.vq_assign_variant(D, K, has_dist=True) == 0
    idx, quant, _ = _C.vq_assign(torch.tensor(x).cuda(), torch.tensor(C).cuda(), want_quant=True, want_dist=False)
    ridx, _, rquant = vs.assign(x, C)
    np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
    np.testing.assert_array_equal(quant.cpu().numpy(), rquant)


def test_prefiltered_assign_agrees_with_the_f32_kernel_on_non_finite_rows():
    """NaN / Inf rows send the group down the plain f32 path: same indices as the f32 kernel (which the distance output selects)."""
    from vqnerf_release_amd import _C
    x, C = _data(1000, 256, 64, seed=3)
    x[5, 7] = np.inf; x[40] = np.nan; x[77, 0] = -np.inf; x[300, 100] = 3e38; x[301] = 1e-30
    xt, Ct = torch.tensor(x).cuda(), torch.tensor(C).cuda()
    a, _, _ = _C.vq_assign(xt, Ct, want_quant=False, want_dist=False)
    b, _, _ = _C.vq_assign(xt, Ct, want_quant=False, want_dist=True)
    np.testing.assert_array_equal(a.cpu().numpy(), b.cpu().numpy())


@pytest.mark.parametrize('N,D,K', [(200000, 256, 15), (200000, 256, 64), (100000, 64, 64), (100000, 252, 64), (50000, 252, 15), (100000, 256, 33)])
def test_assign_matches_fp64_where_gap_is_clear_and_reports_near_ties(N, D, K):
    """The INDEPENDENT check of the index assignment (oracle/vq_strict.c restates the kernel's own summation order): against the float64
    evaluation of vq_layers.py:277-301 wherever the two best distances are more than 1e-5 apart -- the K <= 16 kernel, the prefiltered
    K = 64 / K = 33 kernel, D = 64 and D = 252 (not a multiple of the 64-feature blocks)."""
    from oracle import decomp as od
    from vqnerf_release_amd import _C
    x, C = _data(N, D, K, seed=5 + K + D)
    idx, _, _ = _C.vq_assign(torch.tensor(x).cuda(), torch.tensor(C).cuda(), want_quant=False)
    d64 = od.vq_distances(torch.tensor(x, dtype=torch.float64), torch.tensor(C, dtype=torch.float64)).numpy()
    s = np.sort(d64, 1)
    clear = (s[:, 1] - s[:, 0]) > 1e-5
    assert clear.mean() > 0.95
    assert np.array_equal(idx.cpu().numpy()[clear], d64.argmin(1)[clear])
    print('near-tie fraction (gap <= 1e-5):', 1 - clear.mean(), ' match on near ties:',
          (idx.cpu().numpy()[~clear] == d64.argmin(1)[~clear]).mean() if (~clear).any() else 1.0)


def test_assign_ties_and_dropout_mask():
    from oracle import vq_strict as vs
    from vqnerf_release_amd import _C
    # exact ties: duplicated codes -> lowest index
    x, C = _data(300, 256, 8, seed=9)
    C2 = np.concatenate([C, C], 1)                                   # K=16, code k == code k+8
    idx, _, _ = _C.vq_assign(torch.tensor(x).cuda(), torch.tensor(C2).cuda())
    assert int(idx.max()) < 8
    # dropout mask incl. "everything dropped"
    rng = np.random.default_rng(0)
    for sel in (np.array(rng.uniform(size=16) > 0.5, np.float32), np.zeros(16, np.float32), np.ones(16, np.float32)):
        idx, quant, dist = _C.vq_assign(torch.tensor(x).cuda(), torch.tensor(C2).cuda(), torch.tensor(sel).cuda(),
                                        want_quant=True, want_dist=True)
        ridx, rdist, rquant = vs.assign(x, C2, sel)
        np.testing.assert_array_equal(dist.cpu().numpy(), rdist)
        np.testing.assert_array_equal(idx.cpu().numpy(), ridx)
        np.testing.assert_array_equal(quant.cpu().numpy(), rquant)


@pytest.mark.parametrize('N,D,K', [(5000, 256, 15), (2048, 256, 64), (33, 64, 5), (0, 256, 15), (4099, 128, 20), (1, 256, 15),
                                   (777, 32, 5), (3000, 320, 9)])
def test_ema_stats(N, D, K):
    from oracle import vq_strict as vs
    from vqnerf_release_amd import _C
    x, C = _data(max(N, 1), D, K, seed=3)
    x = x[:N]
    rng = np.random.default_rng(1)
    idx = rng.integers(0, K, N).astype(np.int64)
    if N:
        idx[idx == 2] = 3                                            # code 2 never used
    counts, dw = _C.vq_ema_stats(torch.tensor(x).cuda().reshape(N, D), torch.tensor(idx).cuda(), K)
    rc, rdw = vs.ema_stats(x.reshape(N, D), idx, K)
    np.testing.assert_array_equal(counts.cpu().numpy(), rc)
    np.testing.assert_allclose(dw.cpu().numpy(), rdw, rtol=1e-5, atol=1e-6)


def test_vector_quantizer_ema_module_matches_oracle_over_steps():
    from oracle import decomp as od
    from vqnerf_release_amd.decomp.nerfactor.networks.vq_layers import VectorQuantizerEMA
    D, K, N = 256, 15, 2048
    vq = VectorQuantizerEMA(D, K, commitment_cost=0.1, seed=2).cuda()
    ecs, edw = od.EMA(0.999, (K,)), od.EMA(0.999, (D, K))
    _, C = _data(1, D, K, seed=0)
    C_dev, C_ref = torch.tensor(C).cuda(), torch.tensor(C)
    for step in range(3):
        x, _ = _data(N, D, K, seed=100 + step)
        roll = torch.tensor(np.random.default_rng(step).uniform(size=(1, K)).astype(np.float32))
        thres = None if step == 0 else 0.2
        xin = torch.tensor(x).cuda().requires_grad_(True)
        out = vq(xin, C_dev, is_training=True, thres=thres, roll=roll)
        ecs0, edw0 = copy.deepcopy(ecs), copy.deepcopy(edw)
        th = None if thres is None else torch.tensor(thres)
        ref = od.vq_ema_call(torch.tensor(x), C_ref, ecs, edw, True, thres=th, roll=roll)
        same = out['encoding_indices'].cpu() == ref['encoding_indices']
        assert same.float().mean() > 0.999                           # numpy-BLAS order vs strict order: near ties only
        if not bool(same.all()):
            # a rounding-level tie between two codes on some row: restate the oracle step on the strict-order indices (the device's, bit-
            # checked against oracle/vq_strict.c above) from the saved EMA state, so that update / loss / perplexity are compared on EVERY run
            ecs, edw = ecs0, edw0
            ref = od.vq_ema_call(torch.tensor(x), C_ref, ecs, edw, True, thres=th, roll=roll, idx=out['encoding_indices'].cpu().reshape(-1))
        np.testing.assert_allclose(out['update'].cpu().numpy(), ref['update'].numpy(), rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(out['loss'].item(), ref['loss'].item(), rtol=1e-5)
        np.testing.assert_allclose(out['perplexity'].item(), ref['perplexity'].item(), rtol=1e-5)
        np.testing.assert_allclose(out['distances'].cpu().numpy(), ref['distances'].numpy(), atol=3e-6)
        # straight-through: d quantize / d inputs == identity ; commitment grad = 2 beta (x - q) / numel
        (out['quantize'].sum() + out['loss']).backward()
        q = out['quantize'].detach()
        want = 1.0 + 2 * 0.1 * (xin.detach() - q) / xin.numel()
        np.testing.assert_allclose(xin.grad.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-7)
        C_dev, C_ref = out['update'].detach(), ref['update']          # the caller assigns the update (vq_nfr.py:582-583)
        C_dev = C_dev / C_dev.norm(dim=0, keepdim=True); C_ref = C_ref / C_ref.norm(dim=0, keepdim=True)


def test_full_size_properties():
    """BASELINE-size run (640k rows): size-independent properties instead of the oracle."""
    from vqnerf_release_amd import _C
    N, D, K = 640000, 256, 16
    g = torch.Generator(device='cuda'); g.manual_seed(0)
    x = torch.rand((N, D), device='cuda', generator=g)
    x = x / x.norm(dim=1, keepdim=True)
    C = torch.rand((D, K), device='cuda', generator=g); C = C / C.norm(dim=0, keepdim=True)
    idx, quant, dist = _C.vq_assign(x, C, want_quant=True, want_dist=True)
    assert torch.equal(dist.argmin(1), idx)                           # idx is the argmin of the returned distances
    assert torch.equal(quant, C.t()[idx])                             # quant is exactly the selected column
    idx2, _, _ = _C.vq_assign(quant.contiguous(), C, want_quant=False)
    assert torch.equal(idx2, idx)                                     # idempotence: codes map to themselves
    counts, dw = _C.vq_ema_stats(x, idx, K)
    assert float(counts.sum()) == N
    torch.testing.assert_close(dw.sum(1), x.sum(0), rtol=1e-4, atol=1e-2)   # checksum of checksums
    perm = torch.randperm(N, device='cuda', generator=g)
    idx_p, _, _ = _C.vq_assign(x[perm].contiguous(), C, want_quant=False)
    assert torch.equal(idx_p, idx[perm])                              # row-order independence


def test_ema_stats_is_deterministic_and_ignores_out_of_range_codes():
    """D % 64 == 0, K <= 64 takes the matrix-pipe form (x^T . onehot on v_mfma_f32_16x16x4_f32, fixed-order reductions),
    other K*D <= 4096 shapes the per-wave LDS images: both bit-identical across runs; all agree with the fp64-accumulated
    oracle; rows whose index is outside [0, K) contribute nothing."""
    import torch
    from oracle import vq_strict
    from vqnerf_release_amd import _C
    rng = np.random.default_rng(3)
    for K, D in ((15, 256), (64, 256), (32, 192), (7, 96)):
        x = rng.uniform(0, 1, (50001, D)).astype(np.float32)
        idx = rng.integers(0, K, 50001)
        xt, it = torch.tensor(x).cuda(), torch.tensor(idx).cuda()
        c1, d1 = _C.vq_ema_stats(xt, it, K)
        c2, d2 = _C.vq_ema_stats(xt, it, K)
        rc, rd = vq_strict.ema_stats(x, idx, K)
        np.testing.assert_array_equal(c1.cpu().numpy(), rc)
        np.testing.assert_allclose(d1.cpu().numpy(), rd, rtol=2e-5, atol=1e-3)
        assert torch.equal(d1, d2) and torch.equal(c1, c2)
        bad = idx.copy()
        bad[::7] = K + (np.arange(len(bad[::7])) % 3)               # K, K+1, K+2: inside the padded code tile, outside [0, K)
        bad[3::11] = -1
        keep = (bad >= 0) & (bad < K)
        c3, d3 = _C.vq_ema_stats(xt, torch.tensor(bad).cuda(), K)
        rc3, rd3 = vq_strict.ema_stats(x[keep], bad[keep], K)
        np.testing.assert_array_equal(c3.cpu().numpy(), rc3)
        np.testing.assert_allclose(d3.cpu().numpy(), rd3, rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize('N,D', [(4097, 256), (33, 64), (1, 4), (300000, 256)])
def test_ste_and_commitment_pass(N, D):
    """vqn_vq_ste_loss: x + (q - x) bit-exact against the same torch expression, mean((q - x)^2) against fp64, bit-identical
    across runs; through the module: identity + 2 beta (x - q) / numel gradients (vq_layers.py:302, :327)."""
    from vqnerf_release_amd import _C
    g = torch.Generator(device='cuda'); g.manual_seed(N)
    x = torch.rand((N, D), device='cuda', generator=g)
    q = torch.rand((N, D), device='cuda', generator=g)
    ste, loss = _C.vq_ste_loss(x, q)
    assert torch.equal(ste, x + (q - x))
    want = ((q.double() - x.double()) ** 2).mean()
    assert abs(float(loss) - float(want)) <= 2e-6 * float(want)
    _, loss2 = _C.vq_ste_loss(x, q, want_ste=False)
    assert torch.equal(loss, loss2)


def test_counts_only_path():
    from vqnerf_release_amd import _C
    rng = np.random.default_rng(0)
    for N, K in ((100003, 15), (5, 64), (0, 8)):
        idx = rng.integers(-1, K + 2, N)                              # includes out-of-range codes: ignored
        got = _C.vq_counts(torch.tensor(idx, dtype=torch.int64).cuda(), K).cpu().numpy()
        want = np.array([(idx == k).sum() for k in range(K)], np.float32)
        np.testing.assert_array_equal(got, want)


def test_gradient_trained_vector_quantizer():
    """VectorQuantizer (vq_layers.py:17-171): indices and distances as the EMA layer's, per-row losses, codebook gradient through
    the gathered codes, straight-through gradient to the inputs."""
    from vqnerf_release_amd.decomp.nerfactor.networks.vq_layers import VectorQuantizer
    D, K, N = 64, 9, 500
    x, C = _data(N, D, K, seed=4)
    vq = VectorQuantizer(D, K, commitment_cost=0.25, seed=0).cuda()
    xin = torch.tensor(x).cuda().requires_grad_(True)
    cb = torch.tensor(C).cuda().requires_grad_(True)
    out = vq(xin, cb, is_training=True)
    d_ref = (xin.detach() ** 2).sum(1, keepdim=True) - 2 * xin.detach() @ cb.detach() + (cb.detach() ** 2).sum(0, keepdim=True)
    assert (out['encoding_indices'] == d_ref.argmin(1)).float().mean() > 0.995
    idx = out['encoding_indices']
    q = cb.detach().t()[idx]
    np.testing.assert_allclose(out['qloss'].detach().cpu().numpy(), ((q - xin.detach()) ** 2).mean(-1).cpu().numpy(), rtol=1e-6)
    np.testing.assert_allclose(out['loss'].detach().cpu().numpy(), (1.25 * ((q - xin.detach()) ** 2).mean(-1)).cpu().numpy(), rtol=1e-6)
    assert out['loss'].shape == (N,) and set(out) == {'quantize', 'loss', 'qloss', 'eloss', 'perplexity', 'encodings', 'encoding_indices', 'distances'}
    (out['quantize'].sum() + out['loss'].sum()).backward()
    want_x = 1.0 + 0.25 * 2 * (xin.detach() - q) / D
    np.testing.assert_allclose(xin.grad.cpu().numpy(), want_x.cpu().numpy(), rtol=1e-5, atol=1e-7)
    want_c = torch.zeros(K, D, device='cuda').index_add_(0, idx, 2 * (q - xin.detach()) / D).t()
    np.testing.assert_allclose(cb.grad.cpu().numpy(), want_c.cpu().numpy(), rtol=1e-5, atol=1e-6)
    only = vq(xin.detach(), cb.detach(), False, thres=torch.tensor([1.0] * 4 + [0.0] + [1.0] * 4))
    assert set(only['encoding_indices'].tolist()) == {4}


# ---- round 2: defined-order l2-normalise + the fused inference kernel (normalise + assign + straight-through + loss + usage) ----
@pytest.mark.parametrize('N,D', [(1, 4), (17, 12), (1000, 64), (333, 252), (4097, 256), (50, 512)])
def test_l2_normalize_rows_is_bit_exact_vs_strict_c(N, D):
    from oracle import vq_strict
    from vqnerf_release_amd import _C
    rng = np.random.default_rng(N + D)
    x = rng.uniform(0, 1, (N, D)).astype(np.float32)
    x[0] = 0.0                                                           # the eps floor: 0 / sqrt(1e-6)
    if N > 2:
        x[1] *= 1e-4                                                     # sum x^2 below eps
    y = _C.l2_normalize_rows(torch.tensor(x).cuda())
    np.testing.assert_array_equal(y.cpu().numpy(), vq_strict.l2_normalize(x))
    ref = x.astype(np.float64) / np.sqrt(np.maximum((x.astype(np.float64) ** 2).sum(1, keepdims=True), 1e-6))
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=3e-7, atol=0)


def test_l2_normalize_rows_backward_matches_autograd():
    from vqnerf_release_amd.decomp.nerfactor.networks.vq_layers import l2_normalize_rows
    from vqnerf_release_amd.decomp.nerfactor.util.math import safe_l2_normalize
    torch.manual_seed(0)
    x = torch.rand(64, 256, device='cuda')
    x[0] *= 1e-5                                                         # below the eps floor: gradient g * s only
    g = torch.randn(64, 256, device='cuda')
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    (l2_normalize_rows(a) * g).sum().backward()
    (safe_l2_normalize(b, axis=1) * g).sum().backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize('N,D,K,masked', [(1, 64, 8, False), (17, 256, 15, False), (1000, 256, 15, True), (100003, 256, 64, False),
                                          (5000, 256, 128, True), (777, 128, 33, False)])
def test_fused_quantize_equals_the_three_kernel_sequence(N, D, K, masked):
    """vqn_vq_quantize_rows == vqn_l2_normalize_rows -> vqn_vq_assign -> vqn_vq_ste_loss (+ usage counts): indices and the
    straight-through rows bit for bit, counts exactly, the commitment term to fp32 rounding (its partial sums are grouped per
    workgroup in both forms, but the groups differ); indices also against the strict-order C oracle."""
    from oracle import vq_strict
    from vqnerf_release_amd import _C
    rng = np.random.default_rng(N + K)
    z = rng.uniform(0, 1, (N, D)).astype(np.float32)                      # un-normalised encoder outputs (sigmoid range)
    C = rng.uniform(0, 1, (D, K)).astype(np.float32)
    C /= np.linalg.norm(C, axis=0, keepdims=True)
    sel = (rng.uniform(size=K) > 0.4).astype(np.float32) if masked else None
    if masked:
        sel[0] = 1.0
    zt, Ct = torch.tensor(z).cuda(), torch.tensor(C).cuda()
    st = None if sel is None else torch.tensor(sel).cuda()
    idx_f, ste_f, loss_f, counts_f = _C.vq_quantize_rows(zt, Ct, sel_mask=st)
    zn = _C.l2_normalize_rows(zt)
    idx_s, quant_s, _ = _C.vq_assign(zn, Ct, sel_mask=st, want_quant=True)
    ste_s, loss_s = _C.vq_ste_loss(zn, quant_s)
    counts_s = _C.vq_counts(idx_s, K)
    assert torch.equal(idx_f, idx_s)
    assert torch.equal(ste_f, ste_s)
    assert torch.equal(counts_f, counts_s) and float(counts_f.sum()) == N
    np.testing.assert_allclose(float(loss_f), float(loss_s), rtol=2e-6)
    ridx, _, _ = vq_strict.assign(vq_strict.l2_normalize(z), C, sel=sel, want_dist=False, want_quant=False)
    np.testing.assert_array_equal(idx_f.cpu().numpy(), ridx)
    # no straight-through output requested: same indices, nothing written
    idx_n, ste_n, _, _ = _C.vq_quantize_rows(zt, Ct, sel_mask=st, want_ste=False)
    assert ste_n is None and torch.equal(idx_n, idx_f)


@pytest.mark.parametrize('N,D,K,masked', [(17, 256, 15, False), (4099, 256, 15, True), (3000, 256, 64, False), (2048, 128, 33, True)])
def test_training_quantiser_is_the_four_kernel_sequence(N, D, K, masked):
    """The training path's fused chain (VectorQuantizerEMA.train_from_raw: vqn_vq_quantize_rows_train) against the sequence it replaces
    (l2_normalize_rows -> forward(is_training=True)): indices, straight-through rows, normalised rows (through the EMA statistics and the
    codebook move they feed) bit for bit, loss to fp32 rounding, d / d z to 2e-6 of its largest entry."""
    from vqnerf_release_amd.decomp.nerfactor.networks.vq_layers import VectorQuantizerEMA, l2_normalize_rows
    rng = np.random.default_rng(N + K)
    z0 = torch.tensor(rng.uniform(0, 1, (N, D)).astype(np.float32)).cuda()
    C = rng.uniform(0, 1, (D, K)).astype(np.float32)
    C = torch.tensor(C / np.linalg.norm(C, axis=0, keepdims=True)).cuda()
    thres = ([0.0] * (K - 3) + [1.0, 1.0, 0.5]) if masked else None
    roll = torch.full((1, K), 0.5).cuda() if masked else None
    g = torch.Generator(device='cuda').manual_seed(1)
    w = torch.randn(N, D, device='cuda', generator=g)
    res = {}
    for fused in (True, False):
        layer = VectorQuantizerEMA(embedding_dim=D, num_embeddings=K, commitment_cost=0.25, seed=0).cuda()
        z = z0.clone().requires_grad_(True)
        vq = layer.train_from_raw(z, C, thres=thres, roll=roll) if fused else layer(l2_normalize_rows(z), C, is_training=True, thres=thres, roll=roll)
        ((vq['quantize'] * w).sum() + 3.0 * vq['loss']).backward()
        res[fused] = (vq['encoding_indices'], vq['quantize'].detach(), vq['loss'].detach(), vq['update'], z.grad, layer.ema_dw.hidden.clone())
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3]) and torch.equal(a[5], b[5])
    np.testing.assert_allclose(float(a[2]), float(b[2]), rtol=2e-6)
    assert float((a[4] - b[4]).abs().max()) <= 2e-6 * float(b[4].abs().max())


@pytest.mark.parametrize('K', [15, 64])
def test_model_call_is_bit_identical_with_and_without_the_fused_quantiser(K):
    """vq_nfr.Model.call / fast_render(gen_embed) / vq_test in inference mode: the one-kernel quantiser against the three-kernel
    sequence -- every output tensor bit for bit (the commitment-loss scalar to fp32 rounding)."""
    from oracle import decomp as od
    from tests.decomp_util import make_config, load_oracle_params, make_batch
    from tests.gpu_util import launches
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    p, specs = od.make_model_params(seed=0, K=K)
    model = load_oracle_params(get_model_class('vq_nfr')(make_config(num_embed=K)), p, 'cuda')
    batch = make_batch(od.make_points(5000, seed=4), 'cuda', bg_every=9)
    out = {}
    for fused in (True, False):
        model.fuse_quantise = fused
        with torch.no_grad(), launches() as rec:
            pred, gt, lk, _ = model.call(batch, mode='vali', thres=[0.0] * (K - 2) + [1.0, 1.0], roll=torch.full((1, K), 0.5).cuda())
            emb = model.fast_embed(batch, mode='test')[3]['embed']
            _, _, lkt, _ = model.vq_test(batch, mode='vali')
        assert rec.ran('vqn_vq_quantize_rows') == fused and rec.ran('vqn_vq_ste_loss') == (not fused)
        out[fused] = (pred, lk, emb, lkt)
    (pa, la, ea, ta), (pb, lb, eb, tb) = out[True], out[False]
    for k in pa:
        assert torch.equal(pa[k], pb[k]), k
    assert torch.equal(la['z'], lb['z']) and torch.equal(la['vqrgb'], lb['vqrgb']) and torch.equal(ea, eb)
    assert torch.equal(ta['vqrgb'], tb['vqrgb']) and torch.equal(ta['usage'], tb['usage'])
    np.testing.assert_allclose(float(la['vqloss']), float(lb['vqloss']), rtol=2e-6)
    assert int(pa['embed'].max()) <= K - 2                                   # the two dropped codes are never chosen


@pytest.mark.parametrize('K', [8, 15, 64])
def test_fused_ema_update_matches_the_torch_statement(K):
    """`vqn_vq_ema_update` (both Sonnet moving averages, the Laplace-smoothed cluster sizes and the codebook move, vq_layers.py:304-325,
    one launch) against the torch statement of the same lines over five training calls: state and `update` to 1e-6 relative (the
    K-long sum and the f64 debias are the only places where the two can round differently), counters exact, unused codes keep their
    codebook column."""
    from vqnerf_release_amd.decomp.nerfactor.networks.vq_layers import VectorQuantizerEMA
    D = 256
    rng = np.random.default_rng(K)
    layers = {}
    for fused in (False, True):
        vq = VectorQuantizerEMA(embedding_dim=D, num_embeddings=K, commitment_cost=0.1, seed=0).cuda()
        vq.fuse_ema_update = fused
        layers[fused] = vq
    cb = torch.tensor(rng.uniform(0, 1, (D, K)).astype(np.float32)).cuda()
    cb = cb / cb.norm(dim=0, keepdim=True)
    for step in range(5):
        x = torch.tensor(rng.uniform(0, 1, (500, D)).astype(np.float32)).cuda()
        x = x / x.norm(dim=1, keepdim=True)
        if step == 2:
            x = cb.t()[torch.randint(0, max(K // 2, 1), (500,)).cuda()] * 0.999 + 1e-4      # leaves the upper half of the codes unused
            x = (x / x.norm(dim=1, keepdim=True)).contiguous()
        outs = {f: layers[f](x, cb, is_training=True) for f in (False, True)}
        a, b = outs[False]['update'], outs[True]['update']
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-6, atol=1e-9)
        assert torch.equal(outs[False]['encoding_indices'], outs[True]['encoding_indices'])
        for name in ('ema_cluster_size', 'ema_dw'):
            m0, m1 = getattr(layers[False], name), getattr(layers[True], name)
            assert int(m0.counter) == int(m1.counter) == step + 1
            np.testing.assert_allclose(m1.hidden.cpu().numpy(), m0.hidden.cpu().numpy(), rtol=1e-6, atol=1e-12)
            np.testing.assert_allclose(m1.average.cpu().numpy(), m0.average.cpu().numpy(), rtol=1e-6, atol=1e-12)
        if step == 2 and K > 2:
            unused = torch.bincount(outs[True]['encoding_indices'], minlength=K) == 0
            assert unused.any() and torch.equal(b[:, unused], cb[:, unused])
        # the lazily computed entries are still there for whoever asks
        assert outs[True]['encodings'].shape == (500, K) and float(outs[True]['perplexity']) > 0
        cb = b.clone() / b.norm(dim=0, keepdim=True)


@pytest.mark.parametrize('N,D', [(1, 256), (1000, 256), (37, 64), (5, 1024)])
def test_l2_normalize_rows_backward_kernel_matches_autograd(N, D):
    """vqn_l2_normalize_rows_bwd against autograd through the torch statement of tf.linalg.l2_normalize (util/math.py:63-64),
    rows below the eps clamp included (gradient g s there)."""
    from vqnerf_release_amd import _C
    from vqnerf_release_amd.decomp.nerfactor.util.math import safe_l2_normalize
    g = torch.Generator(device='cuda').manual_seed(N)
    x = torch.randn(N, D, device='cuda', generator=g)
    if N > 2:
        x[1] *= 1e-5                                              # sum x^2 < eps = 1e-6: the clamp is active
        x[2] = 0.0
    gy = torch.randn(N, D, device='cuda', generator=g)
    xr = x.clone().requires_grad_(True)
    safe_l2_normalize(xr, axis=1).backward(gy)
    got = _C.l2_normalize_rows_bwd(x, gy, 1e-6)
    ref = xr.grad
    assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


def test_ste_commitment_backward_kernel_is_the_framework_sequence_bit_for_bit():
    from vqnerf_release_amd import _C
    g = torch.Generator(device='cuda').manual_seed(0)
    x, q, gs = (torch.randn(513, 256, device='cuda', generator=g) for _ in range(3))
    gl = torch.tensor(0.37, device='cuda')
    ref = gs + (x - q) * (gl * (2.0 / x.numel()))
    assert torch.equal(_C.vq_ste_loss_bwd(x, q, gs, gl), ref)
    assert torch.equal(_C.vq_ste_loss_bwd(x, q, None, gl), (x - q) * (gl * (2.0 / x.numel())))
