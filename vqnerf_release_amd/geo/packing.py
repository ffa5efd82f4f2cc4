"""Host-side weight packing for the fused NeuS kernels (csrc/neus_mlp.hip, csrc/mlp_prims.h).

A pack is a pure gather of the effective weight matrix (index tensors are built once per network
shape with numpy; applying them is one `torch.take` per matrix on the device), so it is cheap to
redo after every optimiser step.  Layout, as consumed by `eng::gemm_tiles`:

    pack[out_tile][k_group][lane][j] = M[row = 32*ot + phi(lane & 31)][col = colmap(k_group, j, lane >> 5)]

with  phi(i) = 2*(i & 3) + 8*(i >> 3) + ((i >> 2) & 1)  and, inside a K segment, the feature held by
(row r, component j, half h) being  32*(r >> 2) + 2*(4*(r & 3) + j) + h.
"""
import math

import numpy as np
import torch

MAX_SDF_LAYERS = 12
MAX_COL_LAYERS = 8
SDF_DESC_INTS = 12 + MAX_SDF_LAYERS * 8
COL_DESC_INTS = 16 + MAX_COL_LAYERS * 8


def _phi():
    i = np.arange(32)
    return 2 * (i & 3) + 8 * (i >> 3) + ((i >> 2) & 1)


PHI = _phi()


def seg_features(n_rows):
    """[n_rows, 64, 4] -> local feature index held by (row, lane, j)."""
    r = np.arange(n_rows)[:, None, None]
    lane = np.arange(64)[None, :, None]
    j = np.arange(4)[None, None, :]
    return 32 * (r >> 2) + 2 * (4 * (r & 3) + j) + (lane >> 5)


def gemm_index(n_rows_out, n_cols, segs):
    """Gather index [n_out_tiles, n_groups, 64, 4] into M.flatten() ++ [0].

    segs: list of (n_rows, col_fn) where col_fn maps the local feature index array to a column
    index array (negative = padding)."""
    n_tiles = (n_rows_out + 31) // 32
    zero_slot = n_rows_out * n_cols
    cols = []
    for n_rows, col_fn in segs:
        cols.append(col_fn(seg_features(n_rows)))
    col = np.concatenate(cols, 0)                                    # [ng, 64, 4]
    ot = np.arange(n_tiles)[:, None, None, None]
    lane = np.arange(64)[None, None, :, None]
    row = 32 * ot + PHI[lane & 31]                                   # [nt,1,64,1]
    row = np.broadcast_to(row, (n_tiles, col.shape[0], 64, 4))
    colb = np.broadcast_to(col[None], row.shape)
    idx = np.where((row < n_rows_out) & (colb >= 0), row * n_cols + colb, zero_slot)
    return idx.astype(np.int64)


def bias_index(n_out):
    """[n_tiles, 2, 16] -> bias[32*ot + 2*rho + h] (or the zero slot n_out)."""
    n_tiles = (n_out + 31) // 32
    ot = np.arange(n_tiles)[:, None, None]
    h = np.arange(2)[None, :, None]
    rho = np.arange(16)[None, None, :]
    f = 32 * ot + 2 * rho + h
    return np.where(f < n_out, f, n_out).astype(np.int64)


def rowdot_index(n_out, n_rows, n_cols, col_fn=None):
    """[n_out, n_rows, 2, 4] image of the rows of M [n_out, n_cols] in activation-image order."""
    r = np.arange(n_rows)[:, None, None]
    h = np.arange(2)[None, :, None]
    j = np.arange(4)[None, None, :]
    f = 32 * (r >> 2) + 2 * (4 * (r & 3) + j) + h
    col = f if col_fn is None else col_fn(f)
    o = np.arange(n_out)[:, None, None, None]
    colb = np.broadcast_to(col[None], (n_out,) + col.shape)
    valid = (colb >= 0) & (colb < n_cols)
    return np.where(valid, o * n_cols + colb, n_out * n_cols).astype(np.int64)


# ---- split-precision engine (csrc/mlp_prims_f16s.h): K advances in 16-feature steps = row pairs (hi, lo) ----
def step_features(n_rows):
    """[n_rows/2, 64, 8] -> local feature held by (step, lane, half-slot): 16 sl + 8 (jj >> 2) + 4 h + (jj & 3)."""
    sl = np.arange(n_rows // 2)[:, None, None]
    h = (np.arange(64) >> 5)[None, :, None]
    jj = np.arange(8)[None, None, :]
    return 16 * sl + 8 * (jj >> 2) + 4 * h + (jj & 3)


def emb_rows_for_f16s(n_feats):
    return 2 * ((n_feats + 15) // 16)


def gemm_index_f16s(n_rows_out, n_cols, segs):
    """[n_out_tiles, n_steps (padded to whole 4-step blocks), 64, 8] gather index into M.flatten() ++ [0]: the A operand of
    v_mfma_f32_32x32x16_f16 -- lane (r, h) holds M[32 ot + r][col(step, h, jj)]; segs as in gemm_index."""
    n_tiles = (n_rows_out + 31) // 32
    col = np.concatenate([col_fn(step_features(n_rows)) for n_rows, col_fn in segs], 0)      # [S,64,8]
    pad = (-col.shape[0]) % 4
    if pad:
        col = np.concatenate([col, np.full((pad, 64, 8), -1, col.dtype)], 0)
    row = 32 * np.arange(n_tiles)[:, None, None, None] + (np.arange(64) & 31)[None, None, :, None]
    row = np.broadcast_to(row, (n_tiles,) + col.shape)
    colb = np.broadcast_to(col[None], row.shape)
    return np.where((row < n_rows_out) & (colb >= 0), row * n_cols + colb, n_rows_out * n_cols).astype(np.int64)


def bias_index_f16s(n_out):
    """[n_tiles, 2, 16]: accumulator register reg of lane half h is output row (reg & 3) + 8 (reg >> 2) + 4 h."""
    n_tiles = (n_out + 31) // 32
    ot = np.arange(n_tiles)[:, None, None]
    h = np.arange(2)[None, :, None]
    reg = np.arange(16)[None, None, :]
    f = 32 * ot + (reg & 3) + 8 * (reg >> 2) + 4 * h
    return np.where(f < n_out, f, n_out).astype(np.int64)


def rowdot_index_f16s(n_out, n_rows, n_cols, col_fn=None):
    """[n_out, n_rows/2, 2, 8] f32 image of the rows of M [n_out, n_cols] in split-image order."""
    f = step_features(n_rows)[:, ::32, :]
    col = f if col_fn is None else col_fn(f)
    o = np.arange(n_out)[:, None, None, None]
    colb = np.broadcast_to(col[None], (n_out,) + col.shape)
    valid = (colb >= 0) & (colb < n_cols)
    return np.where(valid, o * n_cols + colb, n_out * n_cols).astype(np.int64)


def split_pack(g):
    """g [T, S, 64, 8] f32 (gemm_index_f16s order) -> flat float32 view of [T, S, 2, 64, 8] f16: hi = f16(w), lo = f16((w - hi) 2^11)."""
    if float(g.abs().max()) > 6.0e4:
        raise ValueError('split-precision packs hold weights as f16 hi/lo: |w| must stay below 6e4')
    hi = g.to(torch.float16)
    lo = ((g - hi.float()) * 2048.0).to(torch.float16)
    return torch.stack([hi, lo], 2).contiguous().view(torch.float32).reshape(-1)


# ---- exact-split engine (csrc/mlp_prims_x3.h): K advances in 16-feature steps = row TRIPLES (p0, p1, p2), 6 rows per 32 features ----
def emb_rows_for_x3(n_feats):
    return 3 * ((n_feats + 15) // 16)


def step_features_n(n_steps):
    """[n_steps, 64, 8] -> local feature held by (step, lane, slot): 16 sl + 8 (jj >> 2) + 4 h + (jj & 3) (as step_features)."""
    sl = np.arange(n_steps)[:, None, None]
    h = (np.arange(64) >> 5)[None, :, None]
    jj = np.arange(8)[None, None, :]
    return 16 * sl + 8 * (jj >> 2) + 4 * h + (jj & 3)


def gemm_index_x3(n_rows_out, n_cols, segs):
    """[n_out_tiles, n_steps (padded to whole 2-step blocks), 64, 8] gather index: the A operand of v_mfma_f32_32x32x16_bf16 before
    the split; segs as in gemm_index with row counts in x3 rows (3 per step)."""
    n_tiles = (n_rows_out + 31) // 32
    for n_rows, _ in segs:
        assert n_rows % 3 == 0
    col = np.concatenate([col_fn(step_features_n(n_rows // 3)) for n_rows, col_fn in segs], 0)
    if col.shape[0] % 2:
        col = np.concatenate([col, np.full((1, 64, 8), -1, col.dtype)], 0)
    row = 32 * np.arange(n_tiles)[:, None, None, None] + (np.arange(64) & 31)[None, None, :, None]
    row = np.broadcast_to(row, (n_tiles,) + col.shape)
    colb = np.broadcast_to(col[None], row.shape)
    return np.where((row < n_rows_out) & (colb >= 0), row * n_cols + colb, n_rows_out * n_cols).astype(np.int64)


def rowdot_index_x3(n_out, n_rows, n_cols, col_fn=None):
    """[n_out, n_rows/3, 2, 8] f32 image of the rows of M [n_out, n_cols] in x3-image order."""
    assert n_rows % 3 == 0
    f = step_features_n(n_rows // 3)[:, ::32, :]
    col = f if col_fn is None else col_fn(f)
    o = np.arange(n_out)[:, None, None, None]
    colb = np.broadcast_to(col[None], (n_out,) + col.shape)
    valid = (colb >= 0) & (colb < n_cols)
    return np.where(valid, o * n_cols + colb, n_out * n_cols).astype(np.int64)


def split3_exact(g):
    """f32 tensor -> (p0, p1, p2) f32 tensors, each with at most 8 significant bits (a bf16 value), p0 + p1 + p2 == g EXACTLY:
    truncation of the low 16 bits of the word, twice on the exact remainders."""
    def trunc(t):
        return (t.contiguous().view(torch.int32) & -65536).view(torch.float32)
    p0 = trunc(g)
    r1 = g - p0
    p1 = trunc(r1)
    p2 = trunc(r1 - p1)
    return p0, p1, p2


def split_pack_x3(g):
    """g [T, S, 64, 8] f32 (gemm_index_x3 order) -> flat float32 view of [T, S, 3, 64, 8] bf16 pieces (no range limit, no scaling)."""
    if not bool(torch.isfinite(g).all()):
        raise ValueError('x3 packs: non-finite weight')
    pieces = [(p.contiguous().view(torch.int32) >> 16).to(torch.int16) for p in split3_exact(g.float())]
    return torch.stack(pieces, 2).contiguous().view(torch.float32).reshape(-1)


def _take(mat, idx_dev):
    flat = torch.cat([mat.reshape(-1), mat.new_zeros(1)])
    return torch.take(flat, idx_dev).reshape(-1)


def wbuf_zero3(like):
    return like.new_zeros(3, dtype=torch.float32)


def emb_rows_for(n_feats):
    return (((n_feats + 1) // 2) + 3) // 4


def ident_cols(n_valid, base=0):
    return lambda f: np.where(f < n_valid, f + base, -1)


_INDEX_FNS = {'f32': (gemm_index, bias_index, rowdot_index),
              'f16s': (gemm_index_f16s, bias_index_f16s, rowdot_index_f16s),
              'x3': (gemm_index_x3, bias_index_f16s, rowdot_index_x3)}      # (the accumulator-order bias image is the f16s one)


class SdfPackPlan:
    """Index tensors + descriptor for an SDFNetwork-shaped MLP (fields.py:9-107)."""

    def __init__(self, dims, skip_in, multires, scale, max_tiles=None, with_reverse=True, mode='f32'):
        # dims: [d0, hidden..., d_out] as in fields.py:24 (d0 = embedded input width)
        assert mode in ('f32', 'f16s', 'x3')
        self.mode = mode                    # 'f16s': packs for csrc/neus_mlp_f16s.hip (f16 pair engine); 'x3': csrc/neus_mlp_x3.hip (exact bf16x3 split)
        self.rpt = 6 if mode == 'x3' else 4 # LDS rows per 32-feature tile
        self.dims = list(dims)
        self.n_lin = len(dims) - 1
        assert 2 <= self.n_lin <= MAX_SDF_LAYERS
        skips = [l for l in skip_in if 0 < l < self.n_lin]
        assert len(skips) <= 1, 'one skip connection supported'
        self.skip = skips[0] if skips else -1
        assert self.skip != self.n_lin - 1, 'skip into the last layer is not supported'
        self.multires = multires
        self.emb = dims[0]
        assert self.emb == 3 + 6 * multires and self.emb <= 64
        self.emb_rows = {'f32': emb_rows_for, 'f16s': emb_rows_for_f16s, 'x3': emb_rows_for_x3}[mode](self.emb)
        self.scale = float(scale)
        # true output width of every linear layer (fields.py:38-41)
        self.out_dims = []
        for l in range(self.n_lin):
            o = dims[l + 1] - dims[0] if (l + 1) == self.skip else dims[l + 1]
            self.out_dims.append(o)
        self.in_dims = [dims[l] for l in range(self.n_lin)]
        self.tiles = [(o + 31) // 32 for o in self.out_dims]
        self.feat_out = self.out_dims[-1] - 1
        self.tiles[-1] = (self.feat_out + 31) // 32 if self.feat_out > 0 else 0
        self.max_tiles = max(max(self.tiles), max_tiles or 1)
        self.with_reverse = with_reverse
        self._build()

    def _build(self):
        E, er = self.emb, self.emb_rows
        gemm_index, bias_index, rowdot_index = _INDEX_FNS[self.mode]
        plan = []          # (kind, layer, index array)
        for l in range(self.n_lin):
            rows_prev = self.rpt * self.tiles[l - 1] if l > 0 else 0
            if l == 0:
                segs = [(er, ident_cols(E))]
            elif l == self.skip:
                prev = self.out_dims[l - 1]
                segs = [(rows_prev, ident_cols(prev)), (er, ident_cols(E, base=prev))]
            else:
                segs = [(rows_prev, ident_cols(self.in_dims[l]))]
            if l < self.n_lin - 1:
                plan.append(('w', l, gemm_index(self.out_dims[l], self.in_dims[l], segs)))
                plan.append(('b', l, bias_index(self.out_dims[l])))
            else:
                if self.feat_out > 0:   # feature rows = rows 1.. of the last layer
                    plan.append(('wfeat', l, gemm_index(self.feat_out, self.in_dims[l], segs)))
                    plan.append(('bfeat', l, bias_index(self.feat_out)))
                plan.append(('wrow', l, rowdot_index(1, rows_prev, self.in_dims[l])))
            if self.with_reverse and l < self.n_lin - 1:
                ksegs = [(self.rpt * self.tiles[l], ident_cols(self.out_dims[l]))]
                if l >= 1:
                    prev = self.out_dims[l - 1]
                    plan.append(('wT', l, gemm_index(prev, self.out_dims[l], ksegs)))
                if l == 0 or l == self.skip:
                    plan.append(('wTE', l, gemm_index(E, self.out_dims[l], ksegs)))
        self.plan = plan
        self._dev_idx = {}

    def _indices(self, device):
        key = str(device)
        if key not in self._dev_idx:
            self._dev_idx[key] = [torch.from_numpy(ix).to(device) for _, _, ix in self.plan]
        return self._dev_idx[key]

    def pack(self, weights, biases):
        """weights[l]: effective [out_l, in_l] (weight-norm already applied), biases[l]: [out_l].
        Returns (wbuf float32 [n], desc int32 numpy [SDF_DESC_INTS])."""
        dev = weights[0].device
        idxs = self._indices(dev)
        chunks, off = [], 0
        layer = [dict(n_out_tiles=self.tiles[l], kA=0, kB=0, w=-1, b=-1, wT=-1, wTE=-1) for l in range(self.n_lin)]
        last_w_off = -1
        for (kind, l, _), ix in zip(self.plan, idxs):
            W = weights[l]
            if l == self.skip:
                W = W / math.sqrt(2.0)
            if kind == 'w':
                src = W
            elif kind == 'b':
                src = biases[l]
            elif kind == 'wfeat':
                src = W[1:]
            elif kind == 'bfeat':
                src = biases[l][1:]
            elif kind == 'wrow':
                src = W[:1]
            elif kind == 'wT':
                src = W[:, :self.out_dims[l - 1]].t()
            elif kind == 'wTE':
                src = W[:, self.out_dims[l - 1]:].t() if l == self.skip else W.t()
            c = _take(src.contiguous(), ix)
            if self.mode != 'f32' and kind in ('w', 'wfeat', 'wT', 'wTE'):
                c = (split_pack if self.mode == 'f16s' else split_pack_x3)(c.reshape(ix.shape))
            assert c.numel() % 4 == 0
            o4 = off // 4
            if kind in ('w', 'wfeat'):
                layer[l]['w'] = o4
            elif kind in ('b', 'bfeat'):
                layer[l]['b'] = o4
            elif kind == 'wrow':
                last_w_off = o4
            elif kind == 'wT':
                layer[l]['wT'] = o4
            elif kind == 'wTE':
                layer[l]['wTE'] = o4
            chunks.append(c)
            off += c.numel()
        # the sdf row's bias rides in the pack (component 0 of one float4): no device -> host copy per re-pack
        last_b_off = off // 4
        chunks.append(torch.cat([biases[self.n_lin - 1][:1].reshape(1).float(), wbuf_zero3(weights[0])]))
        off += 4
        wbuf = torch.cat(chunks).contiguous()
        desc = np.zeros(SDF_DESC_INTS, np.int32)
        desc[0:6] = [self.n_lin, self.skip, self.multires, self.emb, self.emb_rows, self.max_tiles]
        desc[6] = np.float32(self.scale).view(np.int32)
        desc[7] = last_w_off
        desc[9] = last_b_off
        for l in range(self.n_lin):
            d = layer[l]
            desc[12 + 8 * l: 12 + 8 * l + 8] = [d['n_out_tiles'], d['kA'], d['kB'], d['w'], d['b'], d['wT'], d['wTE'], 0]
        return wbuf, desc


class ColPackPlan:
    """RenderingNetwork-shaped MLP (fields.py:111-172); input order [pts, view_embed, normals, feat]."""

    def __init__(self, d_feature, mode, d_hidden, n_layers, d_out, multires_view, squeeze_out, feat_tiles, matrix_mode='f32'):
        assert matrix_mode in ('f32', 'f16s', 'x3')
        self.matrix_mode = matrix_mode
        rpt = 6 if matrix_mode == 'x3' else 4
        gemm_index, bias_index, rowdot_index = _INDEX_FNS[matrix_mode]
        self.mode = mode
        self.n_view = (3 + 6 * multires_view) if mode in ('idr', 'no_normal') else 0
        self.has_normal = 1 if mode in ('idr', 'no_view_dir') else 0
        self.extra = 3 + self.n_view + 3 * self.has_normal
        self.extra_rows = {'f32': emb_rows_for, 'f16s': emb_rows_for_f16s, 'x3': emb_rows_for_x3}[matrix_mode](self.extra)
        self.d_feature = d_feature
        self.dims = [self.extra + d_feature] + [d_hidden] * n_layers + [d_out]
        self.n_lin = len(self.dims) - 1
        assert 2 <= self.n_lin <= MAX_COL_LAYERS and d_out == 3
        self.tiles = [(self.dims[l + 1] + 31) // 32 for l in range(self.n_lin)]
        self.squeeze_out = 1 if squeeze_out else 0
        self.feat_tiles = feat_tiles
        plan = []
        for l in range(self.n_lin - 1):
            if l == 0:
                segs = [(rpt * feat_tiles, ident_cols(d_feature, base=self.extra)), (self.extra_rows, ident_cols(self.extra))]
            else:
                segs = [(rpt * self.tiles[l - 1], ident_cols(self.dims[l]))]
            plan.append(('w', l, gemm_index(self.dims[l + 1], self.dims[l], segs)))
            plan.append(('b', l, bias_index(self.dims[l + 1])))
        L = self.n_lin - 1
        plan.append(('wrow', L, rowdot_index(d_out, rpt * self.tiles[L - 1], self.dims[L])))
        self.plan = plan
        self._dev_idx = {}

    def _indices(self, device):
        key = str(device)
        if key not in self._dev_idx:
            self._dev_idx[key] = [torch.from_numpy(ix).to(device) for _, _, ix in self.plan]
        return self._dev_idx[key]

    def pack(self, weights, biases):
        dev = weights[0].device
        chunks, off = [], 0
        layer = [dict(n_out_tiles=self.tiles[l], w=-1, b=-1) for l in range(self.n_lin)]
        last_w_off = -1
        for (kind, l, _), ix in zip(self.plan, self._indices(dev)):
            src = weights[l] if kind in ('w', 'wrow') else biases[l]
            c = _take(src.contiguous(), ix)
            if self.matrix_mode != 'f32' and kind == 'w':
                c = (split_pack if self.matrix_mode == 'f16s' else split_pack_x3)(c.reshape(ix.shape))
            o4 = off // 4
            if kind == 'w':
                layer[l]['w'] = o4
            elif kind == 'b':
                layer[l]['b'] = o4
            else:
                last_w_off = o4
            chunks.append(c)
            off += c.numel()
        last_b_off = off // 4
        chunks.append(torch.cat([biases[self.n_lin - 1][:3].reshape(3).float(), wbuf_zero3(weights[0])[:1]]))
        off += 4
        wbuf = torch.cat(chunks).contiguous()
        desc = np.zeros(COL_DESC_INTS, np.int32)
        desc[0:8] = [self.n_lin, self.n_view, self.has_normal, self.extra, self.extra_rows, 3, self.squeeze_out, last_w_off]
        desc[12] = last_b_off
        for l in range(self.n_lin):
            d = layer[l]
            desc[16 + 8 * l: 16 + 8 * l + 8] = [d['n_out_tiles'], 0, 0, d['w'], d['b'], -1, -1, 0]
        return wbuf, desc
