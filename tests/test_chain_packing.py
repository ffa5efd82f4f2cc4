"""CPU: the layer programs + weight packs of vqnerf_release_amd/decomp/packing.py, executed by a numpy emulator
of csrc/mlp_chain.hip's data movement (activation image, A-fragment packs, MFMA 32x32x2 operand maps), reproduce the
oracle's Dense stacks.  This pins the host logic (row allocation, gather indices, descriptor fields) without a GPU."""
import numpy as np
import pytest
import torch

from oracle import decomp as od
from vqnerf_release_amd.decomp import packing as pk

PHI = np.array([2 * (i & 3) + 8 * (i >> 3) + ((i >> 2) & 1) for i in range(32)])


def row_feat(r, h, j):
    return 32 * (r >> 2) + 2 * (4 * (r & 3) + j) + h


def act(a, x):
    if a == 1:
        return np.maximum(x, 0)
    if a == 3:
        return 1.0 / (1.0 + np.exp(-x))
    return x


def emulate(desc, wbuf, x, out_widths):
    """x [P<=32, in_stride] -> list of outputs; float64 arithmetic over the packed float32 weights."""
    d = np.asarray(desc)
    w4 = np.asarray(wbuf, np.float64).reshape(-1, 4)
    n_layers, in_mode, in_feats, in_rows, in_row0, n_freqs, total_rows = [int(v) for v in d[:7]]
    P = x.shape[0]
    lds = np.full((total_rows, 64, 4), np.nan)
    if in_mode == 1:
        feats = od.posenc(torch.tensor(x, dtype=torch.float64), n_freqs).numpy()
    else:
        feats = x.astype(np.float64)
    for r in range(in_rows):
        for lane in range(64):
            p, h = lane & 31, lane >> 5
            for j in range(4):
                f = row_feat(r, h, j)
                lds[in_row0 + r, lane, j] = feats[p, f] if (p < P and f < in_feats) else 0.0
    outs = [np.zeros((P, w)) for w in out_widths]
    for l in range(n_layers):
        L = d[16 + 16 * l: 32 + 16 * l]
        kind, a, tiles, kA0, kA, kB0, kB, dst0, w_off, b_off, slot, out_feats = [int(v) for v in L[:12]]
        rows = [kA0 + i for i in range(kA)] + [kB0 + i for i in range(kB)]
        ng = len(rows)
        if kind == 0:
            for ot in range(tiles):
                # D[i][n] = bias + sum_{g,j,h} A[g][lane=(i,h)][j] * B[g][lane=(n,h)][j]
                A = w4[w_off + ot * ng * 64: w_off + (ot + 1) * ng * 64].reshape(ng, 2, 32, 4)      # [g,h,i,j]
                B = np.stack([lds[r] for r in rows]).reshape(ng, 2, 32, 4)                              # [g,h,n,j]
                D = np.einsum('ghij,ghnj->in', A, B)
                bias = w4[b_off + ot * 8: b_off + (ot + 1) * 8].reshape(2, 16)                          # [h][reg]
                for lane in range(64):
                    n_, h = lane & 31, lane >> 5
                    for reg in range(16):
                        i = (reg & 3) + 8 * (reg >> 2) + 4 * h
                        lds[dst0 + ot * 4 + (reg >> 2), lane, reg & 3] = act(a, D[i, n_] + bias[h, reg])
            if slot >= 0:
                for f in range(out_feats):
                    r, fi = f >> 3, f & 7
                    h, j = fi & 1, fi >> 1
                    assert row_feat(r, h, j) == f
                    outs[slot][:, f] = lds[dst0 + r, np.arange(P) + 32 * h, j]
        else:
            img = w4[w_off: w_off + tiles * ng * 2].reshape(tiles, ng, 2, 4)
            B = np.stack([lds[r] for r in rows]).reshape(ng, 2, 32, 4)
            bias4 = d[16 + 16 * l + 12: 16 + 16 * l + 16].view(np.float32).astype(np.float64)
            val = np.einsum('oghj,ghnj->no', img, B)
            outs[slot][:, :tiles] = act(a, val[:P] + bias4[None, :tiles])
    return outs


def _params(names, p):
    out = {}
    for n in names:
        for i, (W, b) in enumerate(p[n]):
            out[f'{n}/{i}'] = (torch.tensor(W), torch.tensor(b))
    return out


def test_encoder_program_matches_oracle():
    p, specs = od.make_model_params(seed=3, K=15)
    fe, bn = specs['fine_enc'], specs['bottleneck']
    b = pk.ChainBuilder('posenc', 63, n_freqs=10)
    y = b.mlp('fine_enc', fe['widths'], fe['act'], fe['skip_at'], b.input)
    b.mlp('bottleneck', bn['widths'], bn['act'], bn['skip_at'], y, out_slot=0, small_last=False)
    plan = b.build()
    assert plan.macs_per_point() == 179968
    assert plan.n_waves == 4 and plan.total_rows <= 72
    wbuf, desc = plan.pack(_params(['fine_enc', 'bottleneck'], p))
    xyz = od.make_points(20, seed=5)['xyz']
    got = emulate(desc, wbuf.numpy(), xyz, [256])[0]
    pt = {k: [(torch.tensor(W, dtype=torch.float64), torch.tensor(bb, dtype=torch.float64)) for W, bb in v]
          for k, v in p.items() if isinstance(v, list)}
    want = od.pred_enc(pt, specs, torch.tensor(xyz, dtype=torch.float64)).numpy()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)


def test_three_heads_share_the_input_rows():
    p, specs = od.make_model_params(seed=4, K=15)
    for fam, widths in (('main', [3, 1, 1]), ('vq', [3, 3, 1])):
        names = [h + '_' + fam for h in ('diff', 'spec', 'rough')]
        b = pk.ChainBuilder('raw', 256)
        for slot, n in enumerate(names):
            s = specs[n]
            b.mlp(n, s['widths'], s['act'], s['skip_at'], b.input, keep=[b.input], out_slot=slot)
        plan = b.build()
        assert plan.macs_per_point() == (296832 if fam == 'main' else 297600)
        assert plan.total_rows * 1024 + 4096 <= 160 * 1024
        wbuf, desc = plan.pack(_params(names, p))
        z = np.random.default_rng(0).uniform(0, 1, (32, 256)).astype(np.float32)
        got = emulate(desc, wbuf.numpy(), z, widths)
        for g, n in zip(got, names):
            pt = [(torch.tensor(W, dtype=torch.float64), torch.tensor(bb, dtype=torch.float64)) for W, bb in p[n]]
            want = od.mlp_forward(pt, specs[n], torch.tensor(z, dtype=torch.float64)).numpy()
            np.testing.assert_allclose(g, want, rtol=0, atol=1e-9)


def test_regions_never_overlap_live_data():
    b = pk.ChainBuilder('raw', 256)
    x = b.input
    y0 = b.dense('a/0', [x], 256, 'relu', keep=[x])
    y1 = b.dense('a/1', [y0], 128, 'relu', keep=[x])
    y2 = b.dense('a/2', [y1, x], 64, 'relu')                   # y0 is dead here: its rows may be reused
    plan = b.build()
    disjoint = lambda a, c: a.row0 >= c.row0 + c.alloc_rows or a.row0 + a.alloc_rows <= c.row0
    assert disjoint(y0, x) and disjoint(y1, x) and disjoint(y1, y0) and disjoint(y2, y1) and disjoint(y2, x)
    assert plan.total_rows == 32 + 32 + 16                       # y2 fits into y0's rows


# ------------------------------------------------------------------ split-precision (f16 hi/lo) programs
def step_feat(sl, h, jj):
    return 16 * sl + 8 * (jj >> 2) + 4 * h + (jj & 3)


def _split(x):
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float64)) * 2048.0).astype(np.float16)
    return hi, lo


def emulate_f16s(desc, wbuf, x, out_widths):
    """numpy emulator of csrc/mlp_chain_f16s.hip: split activation image (row pairs hi / lo), f16 hi/lo A-fragment packs,
    v_mfma_f32_32x32x16_f16 operand maps, three products per step; products and sums in float64."""
    d = np.asarray(desc)
    wb = np.ascontiguousarray(np.asarray(wbuf, np.float32))
    w16 = wb.view(np.float16).astype(np.float64).reshape(-1, 8)          # one 16-byte lane fragment per row
    w4 = wb.astype(np.float64).reshape(-1, 4)
    n_layers, in_mode, in_feats, in_rows, in_row0, n_freqs, total_rows = [int(v) for v in d[:7]]
    assert in_rows % 2 == 0
    P = x.shape[0]
    lds = np.full((total_rows, 64, 8), np.nan)                           # halves, as float64
    feats = od.posenc(torch.tensor(x, dtype=torch.float64), n_freqs).numpy() if in_mode == 1 else x.astype(np.float64)

    def store(row, lane, vals):
        hi, lo = _split(np.asarray(vals, np.float64))
        lds[row, lane], lds[row + 1, lane] = hi, lo

    for sl in range(in_rows // 2):
        for lane in range(64):
            p, h = lane & 31, lane >> 5
            v = [feats[p, step_feat(sl, h, jj)] if (p < P and step_feat(sl, h, jj) < in_feats) else 0.0 for jj in range(8)]
            store(in_row0 + 2 * sl, lane, v)
    outs = [np.zeros((P, w)) for w in out_widths]
    for l in range(n_layers):
        L = d[16 + 16 * l: 32 + 16 * l]
        kind, a, tiles, kA0, kA, kB0, kB, dst0, w_off, b_off, slot, out_feats = [int(v) for v in L[:12]]
        assert kA % 2 == 0 and kB % 2 == 0
        rows = [kA0 + i for i in range(kA)] + [kB0 + i for i in range(kB)]
        nr, ns = len(rows), len(rows) // 2
        B = np.stack([lds[r] for r in rows]).reshape(ns, 2, 2, 32, 8)                                 # [step, part, h, n, jj]
        if kind == 0:
            for ot in range(tiles):
                nrp = 8 * ((nr + 7) // 8)                                                               # packs hold whole 4-step blocks
                A = w16[w_off + ot * nrp * 64: w_off + (ot + 1) * nrp * 64].reshape(nrp // 2, 2, 2, 32, 8)
                assert not A[ns:].any()                                                                # zero padding
                A = A[:ns]                                                                              # [step, part, h, i, jj]
                acc1 = np.einsum('shij,shnj->in', A[:, 0], B[:, 0])
                acc2 = np.einsum('shij,shnj->in', A[:, 0], B[:, 1]) + np.einsum('shij,shnj->in', A[:, 1], B[:, 0])
                bias = w4[b_off + ot * 8: b_off + (ot + 1) * 8].reshape(2, 16)
                for lane in range(64):
                    n_, h = lane & 31, lane >> 5
                    for s in range(2):
                        v = []
                        for jj in range(8):
                            reg = 8 * s + jj
                            i = (reg & 3) + 8 * (reg >> 2) + 4 * h
                            assert i == step_feat(s, h, jj)                                           # registers ARE the half-slots
                            v.append(act(a, acc1[i, n_] + bias[h, reg] + acc2[i, n_] / 2048.0))
                        store(dst0 + ot * 4 + 2 * s, lane, v)
                        if slot >= 0 and n_ < P:
                            for jj in range(8):
                                f = 32 * ot + step_feat(s, h, jj)
                                if f < out_feats:
                                    outs[slot][n_, f] = v[jj]
        elif kind == 1:
            img = w4[w_off: w_off + tiles * ns * 2 * 2].reshape(tiles, ns, 2, 8)
            X = B[:, 0] + B[:, 1] / 2048.0                                                             # [step, h, n, jj]
            bias4 = d[16 + 16 * l + 12: 16 + 16 * l + 16].view(np.float32).astype(np.float64)
            val = np.einsum('oshj,shnj->no', img, X)
            outs[slot][:, :tiles] = act(a, val[:P] + bias4[None, :tiles])
    return outs


def test_split_precision_programs_match_oracle():
    p, specs = od.make_model_params(seed=3, K=15)
    fe, bn = specs['fine_enc'], specs['bottleneck']
    b = pk.ChainBuilder('posenc', 63, n_freqs=10, mode='f16s')
    y = b.mlp('fine_enc', fe['widths'], fe['act'], fe['skip_at'], b.input)
    b.mlp('bottleneck', bn['widths'], bn['act'], bn['skip_at'], y, out_slot=0, small_last=False)
    plan = b.build()
    assert plan.macs_per_point() == 179968 and plan.total_rows <= 72
    wbuf, desc = plan.pack(_params(['fine_enc', 'bottleneck'], p))
    xyz = od.make_points(20, seed=5)['xyz']
    got = emulate_f16s(desc, wbuf.numpy(), xyz, [256])[0]
    pt = {k: [(torch.tensor(W, dtype=torch.float64), torch.tensor(bb, dtype=torch.float64)) for W, bb in v]
          for k, v in p.items() if isinstance(v, list)}
    want = od.pred_enc(pt, specs, torch.tensor(xyz, dtype=torch.float64)).numpy()
    # hi + lo carries ~22 bits, the lo*lo products are dropped: ~1e-6 of the layer's magnitude, not bitwise
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-6 * np.abs(want).max())
    assert np.abs(got - want).max() > 0                              # (and it is not the f32 path in disguise)

    names = ['diff_vq', 'spec_vq', 'rough_vq']
    b = pk.ChainBuilder('raw', 256, mode='f16s')
    for slot, n in enumerate(names):
        s = specs[n]
        b.mlp(n, s['widths'], s['act'], s['skip_at'], b.input, keep=[b.input], out_slot=slot)
    plan = b.build()
    wbuf, desc = plan.pack(_params(names, p))
    z = np.random.default_rng(0).uniform(0, 1, (32, 256)).astype(np.float32)
    got = emulate_f16s(desc, wbuf.numpy(), z, [3, 3, 1])
    for g, n in zip(got, names):
        ptn = [(torch.tensor(W, dtype=torch.float64), torch.tensor(bb, dtype=torch.float64)) for W, bb in p[n]]
        want = od.mlp_forward(ptn, specs[n], torch.tensor(z, dtype=torch.float64)).numpy()
        np.testing.assert_allclose(g, want, rtol=0, atol=3e-6 * max(1.0, np.abs(want).max()))
    with pytest.raises(ValueError):
        pk.split_pack(torch.full((1, 1, 64, 8), 7.0e4))


# ------------------------------------------------------------------ NeuS pack plans in split-precision mode
def test_sdf_pack_plan_f16s_forward_matches_oracle():
    """SdfPackPlan(mode='f16s'): the forward packs (f16 hi/lo A fragments in whole 64-feature blocks, register-order biases, the
    skip layer's two K segments, the sdf row image) run through a numpy model of csrc/neus_mlp_f16s.hip's forward reproduce
    oracle.geo.sdf_only to the split-precision tolerance."""
    from oracle import geo as og
    from vqnerf_release_amd.geo import packing as gp
    cfg = og.SMALL_CFG
    c = cfg['sdf']
    plan = gp.SdfPackPlan(og.sdf_dims(cfg), c['skip_in'], c['multires'], c['scale'], mode='f16s')
    p = og.to_torch(og.make_sdf_params(cfg, 0))
    W = [og.wn_weight(p, l) for l in range(plan.n_lin)]
    b = [p[f'lin{l}.bias'] for l in range(plan.n_lin)]
    wbuf, desc = plan.pack(W, b)
    wb = np.ascontiguousarray(wbuf.numpy())
    w16 = wb.view(np.float16).astype(np.float64).reshape(-1, 8)
    w4 = wb.astype(np.float64).reshape(-1, 4)
    n_lin, skip, multires, emb, emb_rows = [int(v) for v in desc[:5]]
    assert emb_rows % 2 == 0 and emb_rows == 2 * ((emb + 15) // 16)
    P = 20
    pts = np.random.default_rng(2).uniform(-1, 1, (P, 3))
    E = od.posenc(torch.tensor(pts * c['scale'], dtype=torch.float64), multires).numpy()          # [P, emb]

    def image(feats, n_rows):
        """[n_steps, part, h, n, jj] halves of a region holding `feats` [P, F]."""
        ns = n_rows // 2
        out = np.zeros((ns, 2, 2, 32, 8))
        for sl in range(ns):
            for h in range(2):
                for jj in range(8):
                    f = step_feat(sl, h, jj)
                    if f < feats.shape[1]:
                        hi, lo = _split(feats[:, f])
                        out[sl, 0, h, :P, jj], out[sl, 1, h, :P, jj] = hi, lo
        return out

    emb_img = image(E, emb_rows)
    cur = None
    for l in range(n_lin - 1):
        tiles, _, _, w_off, b_off = [int(v) for v in desc[12 + 8 * l: 12 + 8 * l + 5]]
        B = emb_img if l == 0 else (np.concatenate([cur, emb_img], 0) if l == skip else cur)
        ns = B.shape[0]
        nbr = 8 * ((2 * ns + 7) // 8)                                                            # pack rows per tile: whole blocks
        out = np.zeros((P, 32 * tiles))
        for ot in range(tiles):
            A = w16[w_off + ot * nbr * 64: w_off + (ot + 1) * nbr * 64].reshape(nbr // 2, 2, 2, 32, 8)
            assert not A[ns:].any()
            A = A[:ns]
            acc = np.einsum('shij,shnj->in', A[:, 0], B[:, 0]) + (np.einsum('shij,shnj->in', A[:, 0], B[:, 1]) +
                                                                 np.einsum('shij,shnj->in', A[:, 1], B[:, 0])) / 2048.0
            bias = w4[b_off + ot * 8: b_off + (ot + 1) * 8].reshape(2, 16)
            for h in range(2):
                for reg in range(16):
                    i = (reg & 3) + 8 * (reg >> 2) + 4 * h
                    x = acc[i, :P] + bias[h, reg]
                    out[:, 32 * ot + i] = np.where(100 * x > 20, x, np.log1p(np.exp(np.minimum(100 * x, 50))) / 100)
        cur = image(out[:, :32 * tiles], 4 * tiles)
    # sdf row: f32 image [1][n_steps][2][8]
    last_w = int(desc[7])
    ns = cur.shape[0]
    img = w4[last_w: last_w + ns * 4].reshape(ns, 2, 8)
    X = cur[:, 0] + cur[:, 1] / 2048.0                                                           # [step, h, n, jj]
    sdf = np.einsum('shj,shnj->n', img, X)[:P] + float(w4[int(desc[9])][0])
    with torch.no_grad():
        want = og.sdf_only({k: v.double() for k, v in p.items()}, cfg, torch.tensor(pts, dtype=torch.float64))[:, 0].numpy()
    np.testing.assert_allclose(sdf / c['scale'], want, rtol=0, atol=5e-6)
