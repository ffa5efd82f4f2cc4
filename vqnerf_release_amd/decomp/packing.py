"""Host-side layer programs + weight packs for the generic fused Dense-stack kernel (csrc/mlp_chain.hip,
descriptor layout include/vqn_chain_desc.h).  Activation-image / A-fragment layouts are those of geo/packing.py.

`ChainBuilder(mode='f16s')` builds the same programs for the split-precision kernel (csrc/mlp_chain_f16s.hip, layouts in
csrc/mlp_prims_f16s.h): K segments advance in 16-feature steps (row pairs hi / lo), weights are packed as f16 hi / lo
fragments (same bytes, same offsets), biases in accumulator-register order.

A program is built once per network shape with `ChainBuilder`; `ChainPlan.pack(params)` then gathers the
current Keras-layout weights (`kernel [in, out]`, `bias [out]`) into one flat device buffer -- a pure index
gather, cheap enough to redo after every optimiser step.
"""
import numpy as np
import torch

from vqnerf_release_amd.geo import packing as geo_packing
from vqnerf_release_amd.geo.packing import gemm_index, bias_index, ident_cols, _take

MAX_LAYERS = 16
MAX_OUTS = 4
LAYER_INTS = 16
DESC_INTS = 16 + MAX_LAYERS * LAYER_INTS
ACT = {None: 0, 'none': 0, 'relu': 1, 'softplus100': 2, 'sigmoid': 3}
LATE_EPILOGUE = 0x100          # ChainLayer.act bit 8: barrier between the K loops and the write-back (in-place output)


class Region:
    """`feats` features of 32 points held in LDS rows [row0, row0 + rows) (8 features per row); a GEMM output
    occupies whole 32-feature tiles (`alloc_rows`), of which only the first `rows` need to be read as K."""

    def __init__(self, row0, feats, alloc_rows=None, mode='f32'):
        self.row0, self.feats = row0, feats
        self.rows = (feats + 7) // 8 if mode == 'f32' else 2 * ((feats + 15) // 16)
        self.alloc_rows = alloc_rows if alloc_rows is not None else self.rows

    @property
    def end(self):
        return self.row0 + self.rows


def _rowdot_index_segs(n_out, n_cols, segs):
    """[n_out, sum(rows), 2, 4] gather index into M[n_out, n_cols].flatten() ++ [0]; segs = [(rows, feats, col_base)]."""
    cols = []
    for rows, feats, base in segs:
        r = np.arange(rows)[:, None, None]
        h = np.arange(2)[None, :, None]
        j = np.arange(4)[None, None, :]
        f = 32 * (r >> 2) + 2 * (4 * (r & 3) + j) + h
        cols.append(np.where(f < feats, f + base, -1))
    col = np.concatenate(cols, 0)                                     # [R,2,4]
    o = np.arange(n_out)[:, None, None, None]
    colb = np.broadcast_to(col[None], (n_out,) + col.shape)
    return np.where(colb >= 0, o * n_cols + colb, n_out * n_cols).astype(np.int64)


def _step_feat(n_rows):
    return geo_packing.step_features(n_rows)


def gemm_index_f16s(n_out, n_cols, segs):
    """segs = [(rows, feats, col_base)] -> geo.packing.gemm_index_f16s (A operand of v_mfma_f32_32x32x16_f16, whole 4-step blocks)."""
    return geo_packing.gemm_index_f16s(n_out, n_cols, [(rows, ident_cols(feats, base=base)) for rows, feats, base in segs])


bias_index_f16s = geo_packing.bias_index_f16s
split_pack = geo_packing.split_pack


def _rowdot_index_segs_f16s(n_out, n_cols, segs):
    """[n_out, n_steps, 2, 8] f32 image of M's rows in split-image order."""
    cols = []
    for rows, feats, base in segs:
        f = _step_feat(rows)[:, ::32, :]                              # lanes 0 and 32 -> h = 0, 1
        cols.append(np.where(f < feats, f + base, -1))
    col = np.concatenate(cols, 0)                                     # [S,2,8]
    o = np.arange(n_out)[:, None, None, None]
    colb = np.broadcast_to(col[None], (n_out,) + col.shape)
    return np.where(colb >= 0, o * n_cols + colb, n_out * n_cols).astype(np.int64)


class ChainBuilder:
    """Builds the layer program; LDS rows are assigned first-fit against the regions still needed."""

    def __init__(self, in_mode, in_feats, n_freqs=0, in_stride=None, mode='f32'):
        assert in_mode in ('raw', 'posenc') and mode in ('f32', 'f16s')
        self.mode = mode
        if in_mode == 'posenc':
            assert in_feats == 3 + 6 * n_freqs
        self.in_mode, self.in_feats, self.n_freqs = in_mode, in_feats, n_freqs
        self.in_stride = in_stride if in_stride is not None else (3 if in_mode == 'posenc' else in_feats)
        self.input = Region(0, in_feats, mode=mode)
        self.layers = []           # dicts
        self.total_rows = self.input.rows
        self.n_slots = 0

    def dense(self, key, segs, out_feats, act, keep=(), out_slot=None, over=None):
        """GEMM layer: K = concat of `segs` (1 or 2 Regions, in Keras concat order), -> new Region.
        `keep`: regions that later layers still read (must not be overwritten).  Rows are assigned in build().
        `over` (one of `segs`): write the output IN PLACE over that K segment -- the kernel then holds every wave's accumulators
        across a workgroup barrier between the K loops and the write-back (one output tile per wave at most: out_feats <= 128
        with 4 waves).  Saves the rows of one region, which is what lets the head programs keep their input resident."""
        assert 1 <= len(segs) <= 2
        tiles = (out_feats + 31) // 32
        dst = Region(None, out_feats, alloc_rows=4 * tiles, mode=self.mode)
        if over is not None:
            assert any(over is s_ for s_ in segs) and tiles <= 4 and 4 * tiles <= over.alloc_rows and self.mode == 'f32'
        self.layers.append(dict(kind=0, key=key, segs=list(segs), out=out_feats, act=ACT[act] | (LATE_EPILOGUE if over is not None else 0),
                                dst=dst, tiles=tiles, over=over,
                                live=[s_ for s_ in segs if s_ is not over] + list(keep), out_slot=-1 if out_slot is None else out_slot))
        if out_slot is not None:
            self.n_slots = max(self.n_slots, out_slot + 1)
        return dst

    def reload_input(self, keep=()):
        """The input image again, in fresh rows: lets a program drop the input while wide activations are live and fetch it
        back (from L2) where a later layer concatenates it."""
        r = Region(None, self.in_feats, mode=self.mode)
        self.layers.append(dict(kind=2, key=None, segs=[], out=0, act=0, dst=r, tiles=0, live=list(keep), out_slot=-1))
        return r

    def dense_small(self, key, segs, n_out, act, out_slot):
        assert 1 <= n_out <= 4 and 1 <= len(segs) <= 2
        self.layers.append(dict(kind=1, key=key, segs=list(segs), out=n_out, act=ACT[act], dst=None, tiles=n_out,
                                out_slot=out_slot))
        self.n_slots = max(self.n_slots, out_slot + 1)

    def mlp(self, prefix, widths, acts, skip_at, x, keep=(), out_slot=None, small_last=True):
        """networks/mlp.py:24-50: Dense chain; after layer i in skip_at the layer output is concat(y, x)."""
        h_segs = [x]
        n = len(widths)
        for i, (w, a) in enumerate(zip(widths, acts)):
            last = i == n - 1
            later_skip = bool(skip_at) and any(s >= i for s in skip_at if s < n - 1)
            kp = list(keep) + ([x] if later_skip else [])
            if last and small_last and w <= 4 and out_slot is not None:
                assert not (skip_at and i in skip_at)
                self.dense_small(f'{prefix}/{i}', h_segs, w, a, out_slot)
                return None
            y = self.dense(f'{prefix}/{i}', h_segs, w, a, keep=kp, out_slot=out_slot if last else None)
            h_segs = [y, x] if (skip_at and i in skip_at) else [y]
        return y if len(h_segs) == 1 else h_segs

    def _assign_rows(self):
        """Place every GEMM output so that it overlaps none of its layer's live regions, minimising the LDS rows
        used (small depth-first search over 'row 0 or right after an already placed region')."""
        gemm = [L for L in self.layers if L['kind'] in (0, 2)]
        best = {'rows': None, 'pos': None}
        placed = [self.input]

        def rec(i, top, pos):
            if best['rows'] is not None and top >= best['rows']:
                return
            if i == len(gemm):
                best['rows'], best['pos'] = top, list(pos)
                return
            L = gemm[i]
            need = L['dst'].alloc_rows
            cands = sorted({0} | {r.row0 + r.alloc_rows for r in placed})
            if L.get('over') is not None:
                cands = [L['over'].row0]                       # in place: over one of its own K segments
            for c in cands:
                if any(c < r.row0 + r.alloc_rows and r.row0 < c + need for r in L['live']):
                    continue
                L['dst'].row0 = c
                placed.append(L['dst'])
                rec(i + 1, max(top, c + need), pos + [c])
                placed.pop()
                L['dst'].row0 = None

        self.input.row0 = 0
        rec(0, self.input.alloc_rows, [])
        assert best['pos'] is not None
        for L, c in zip(gemm, best['pos']):
            L['dst'].row0 = c
        self.total_rows = best['rows']

    def build(self):
        assert len(self.layers) <= MAX_LAYERS and self.n_slots <= MAX_OUTS
        self._assign_rows()
        return ChainPlan(self)


class ChainPlan:
    def __init__(self, b):
        self.b = b
        self.layers = b.layers
        self.total_rows = b.total_rows
        # <= 4-output layers keep their weight images in LDS (16 B x 2 lane halves per K row and output)
        self.small_w4, off4 = 0, 0
        for L in self.layers:
            if L['kind'] == 1:
                L['lds_w_off'] = off4
                off4 += L['out'] * sum(s.rows for s in L['segs']) * 2
        self.small_w4 = off4
        lds = self.total_rows * 1024 + 8 * 32 * 4 * 4 + 16 * self.small_w4
        assert lds <= 160 * 1024, f'program needs {lds} B of LDS'
        self.n_waves = 4 if 2 * lds <= 160 * 1024 else 8
        self.gather = []
        for L in self.layers:
            in_feats = sum(s.feats for s in L['segs'])
            if L['kind'] == 2:
                L['k_rows'], L['in_feats'] = [], 0
                self.gather.append((None, None))
                continue
            if L['kind'] == 0 and b.mode == 'f32':
                segs, base = [], 0
                for s in L['segs']:
                    segs.append((s.rows, ident_cols(s.feats, base=base)))
                    base += s.feats
                L['k_rows'] = [sg[0] for sg in segs]
                self.gather.append((gemm_index(L['out'], in_feats, segs), bias_index(L['out'])))
            else:
                segs, base = [], 0
                for s in L['segs']:
                    segs.append((s.rows, s.feats, base))
                    base += s.feats
                L['k_rows'] = [sg[0] for sg in segs]
                if L['kind'] == 0:
                    self.gather.append((gemm_index_f16s(L['out'], in_feats, segs), bias_index_f16s(L['out'])))
                elif b.mode == 'f16s':
                    self.gather.append((_rowdot_index_segs_f16s(L['out'], in_feats, segs), None))
                else:
                    self.gather.append((_rowdot_index_segs(L['out'], in_feats, segs), None))
            L['in_feats'] = in_feats
        self._dev = {}

    def _indices(self, device):
        k = str(device)
        if k not in self._dev:
            self._dev[k] = [(None if w is None else torch.from_numpy(w).to(device), None if bi is None else torch.from_numpy(bi).to(device))
                            for w, bi in self.gather]
        return self._dev[k]

    def macs_per_point(self):
        return sum(L['in_feats'] * L['out'] for L in self.layers)

    def pack(self, params):
        """params[key] = (kernel [in, out], bias [out]) device tensors.  -> (wbuf, desc int32 numpy)."""
        b = self.b
        dev = params[next(L['key'] for L in self.layers if L['kind'] != 2)][0].device
        chunks, off = [], 0
        desc = np.zeros(DESC_INTS, np.int32)
        desc[0:10] = [len(self.layers), 1 if b.in_mode == 'posenc' else 0, b.in_feats, b.input.rows, b.input.row0,
                      b.n_freqs, self.total_rows, self.n_waves, b.in_stride, self.small_w4]
        small_bias = []
        for li, (L, (wi, bi)) in enumerate(zip(self.layers, self._indices(dev))):
            if L['kind'] == 2:
                base = 16 + LAYER_INTS * li
                desc[base:base + 12] = [2, 0, 0, 0, 0, 0, 0, L['dst'].row0, 0, -1, -1, 0]
                continue
            W, bias = params[L['key']]
            assert tuple(W.shape) == (L['in_feats'], L['out']), (L['key'], tuple(W.shape), (L['in_feats'], L['out']))
            M = W.t().contiguous()                                  # [out, in]
            c = _take(M, wi)
            if b.mode == 'f16s' and L['kind'] == 0:
                c = split_pack(c.reshape(wi.shape))
            w_off = off // 4
            chunks.append(c); off += c.numel()
            b_off = -1
            if bi is not None:
                cb = _take(bias.contiguous(), bi)
                b_off = off // 4
                chunks.append(cb); off += cb.numel()
            segs = L['segs']
            kA0, kA = segs[0].row0, L['k_rows'][0]
            kB0, kB = (segs[1].row0, L['k_rows'][1]) if len(segs) == 2 else (0, 0)
            dst0 = L['dst'].row0 if L['dst'] is not None else L['lds_w_off']
            base = 16 + LAYER_INTS * li
            desc[base:base + 12] = [L['kind'], L['act'], L['tiles'], kA0, kA, kB0, kB, dst0, w_off, b_off, L['out_slot'],
                                    L['out'] if L['kind'] == 0 else 0]
            if L['kind'] == 1:
                small_bias.append((base + 12, bias))
        if small_bias:                                               # one small D2H copy per re-pack
            flat = torch.cat([bb.reshape(-1).float() for _, bb in small_bias]).detach().cpu().numpy()
            o = 0
            for pos, bb in small_bias:
                n = bb.numel()
                desc[pos:pos + n] = flat[o:o + n].astype(np.float32).view(np.int32)
                o += n
        return torch.cat(chunks).contiguous(), desc
