"""Training passes of the reflectance Dense stacks on the exact-split engine (csrc/refl_train_x3.hip) -- round 4.

What the reference gets from `tape.gradient` through `_pred_enc_at` and the three `_pred_*_at` heads (vq_nfr.py:771-828, nfr_unit.py:329-391
over networks/mlp.py:24-50) under train_nfr.py:562-576, as TWO launches per stack and direction instead of interpreted tile programs:

  ReflStackEngine(enc_nets=[fine_enc, bottleneck], heads=[diff, spec, rough])   xyz -> posenc -> ... -> z -> three heads   ("A")
  ReflStackEngine(enc_nets=None,                   heads=[diff_vq, spec_vq, rough_vq])   z_vq rows -> three heads          ("B")

`forward` leaves every layer's output in the tile format the weight-gradient contraction reads, `backward` every layer's per-point
adjoint; the contractions are the batched ones of geo/train_programs.py (WgradBatch).  Weights keep the Keras layout (kernel [in, out]).
Packs: ONE gather + exact three-way split launch (vqn_pack_x3_gather) for all GEMM matrices of both directions and one gather of the
thin f32 images, from the flat parameter vector; the index arrays and the descriptor do not depend on the weight values (cached).
Round 5: the stage-3 stack (ref_nfr.py:137-152,203-213) -- rgb_enc (3 raw features -> 256 -> 256 -> 256) + the diffuse / roughness heads over
[z_xyz ; z_ref] (512 wide) -- runs here too: `ReflStackEngine(enc_nets=[rgb_enc], n_freqs=0, heads=[diff_out, rough_out], zx=True)`.  z_xyz (the
frozen stage-2 encoder's output) is the heads' SECOND input (`zx`): a second image region in the forward kernel, two K segments in the
heads' first GEMM, a third row-dot share in their last layer; the backward kernel is the same (no adjoint flows into z_xyz), the weight
gradients of the Dense kernels' z_xyz rows are two more contractions per head.  Stacks this engine does not cover keep
decomp/train_programs.py's interpreter."""
import ctypes

import numpy as np
import torch

from vqnerf_release_amd import _C
from vqnerf_release_amd.geo import packing
from vqnerf_release_amd.geo.train_programs import FlatLayout, WgradBatch, _ident

ACTS = {None: 0, 'relu': 1, 'sigmoid': 3}
RT_MAX_L, RT_MAX_H = 8, 3
DESC_INTS = 16 + 5 * RT_MAX_L + 14 * RT_MAX_H


def _tl(f):
    return (f + 31) // 32


_CUS = {}


def _num_cus(device):
    k = str(device)
    if k not in _CUS:
        _CUS[k] = int(torch.cuda.get_device_properties(device).multi_processor_count) if torch.device(device).type == 'cuda' else 256
    return _CUS[k]


def _tiles_per_block():
    """point tiles per partial block of the contractions at small batches (VQN_WGRAD_TPB; 1 = one block per tile, rounds 1-4)"""
    import os
    return int(os.environ.get('VQN_WGRAD_TPB', '2'))


class ReflStackEngine:
    n_split = 256

    @staticmethod
    def supports(enc_nets, heads, z_dim, emb_feats=0, zx=False):
        """the shapes csrc/refl_train_x3.hip runs: layers of at most 256 outputs, at most one skip-concat (of the encoding), standard heads;
        zx: the heads read [zx ; z] (2 z_dim wide: their first kernel has 2 z_dim rows, their last w1 + 2 z_dim)"""
        if z_dim > 256 or z_dim % 4 or len(heads) > RT_MAX_H:
            return False
        if zx and (not heads or z_dim % 32):
            return False
        for net in heads:
            if not (len(net.widths) == 3 and net.skip_at == [1] and net.act == ['relu', 'relu', 'sigmoid'] and net.widths[0] <= 256
                    and net.widths[1] <= 256 and 1 <= net.widths[2] <= 3):
                return False
        if enc_nets:
            if not (3 <= emb_feats <= 64):
                return False
            n, skips = 0, 0
            for ni, net in enumerate(enc_nets):
                # a skip-concat AFTER a net's last layer would hand [y ; x] to the next net's first layer: not a shape the kernel runs
                if net.skip_at is not None and any(int(si) >= len(net.widths) - 1 for si in net.skip_at):
                    return False
                for li, (w, a) in enumerate(zip(net.widths, net.act)):
                    if w > 256 or a not in ACTS:
                        return False
                    if net.skip_at is not None and (li - 1) in net.skip_at:
                        skips += 1
                        if ni != 0 or n == 0:
                            return False
                    n += 1
            if n > RT_MAX_L or skips > 1 or enc_nets[-1].widths[-1] != z_dim:
                return False
        return bool(enc_nets) or bool(heads)

    def __init__(self, enc_nets, n_freqs, heads, z_dim, device, zx=False):
        self.device, self.Z, self.heads = device, z_dim, list(heads)
        self.zx = bool(zx)                 # heads over [zx ; z]: a second input of z_dim features in FRONT of z (the order of ref_nfr's concat)
        self.enc_nets = list(enc_nets) if enc_nets else []
        self.E = 3 + 6 * n_freqs if self.enc_nets else 0
        self.layers, d_prev = [], self.E
        for ni, net in enumerate(self.enc_nets):
            for li, (w, a) in enumerate(zip(net.widths, net.act)):
                skip_in = net.skip_at is not None and (li - 1) in net.skip_at
                self.layers.append(dict(in_y=d_prev, out=w, act=ACTS[a], skip=skip_in))
                d_prev = w
        self.nE, self.nH = len(self.layers), len(self.heads)
        assert self.supports(self.enc_nets, self.heads, z_dim, self.E, zx=self.zx)
        self._dev = None

    # ------------------------------------------------------------------ parameters
    def params(self):
        """[kernel_0, bias_0, kernel_1, ...] of the encoder layers, then of every head's three layers -- the order of forward()'s `params`"""
        ps = []
        for net in self.enc_nets + self.heads:
            for layer in net.layers:
                ps += [layer.kernel, layer.bias]
        return ps

    def layout(self):
        shp = []
        for l, L in enumerate(self.layers):
            shp += [('W%d' % l, (L['in_y'] + (self.E if L['skip'] else 0), L['out'])), ('b%d' % l, (L['out'],))]
        for k, net in enumerate(self.heads):
            w0, w1, c = net.widths
            zin = self.Z * (2 if self.zx else 1)
            shp += [('H%d_W0' % k, (zin, w0)), ('H%d_b0' % k, (w0,)), ('H%d_W1' % k, (w0, w1)), ('H%d_b1' % k, (w1,)),
                    ('H%d_W2' % k, (w1 + zin, c)), ('H%d_b2' % k, (c,))]
        return FlatLayout(shp)

    def _static(self):
        """(layout, int32 gather index of the piece pack, its K-step count, int64 gather index of the f32 images, descriptor)"""
        if self._dev is not None:
            return self._dev
        L = self.layout()
        chunks, off, fch, foff = [], [0], [], [0]

        def add(view, idx):                                   # -> float4 offset inside the PIECE pack (192 float4 per tile and K step)
            src = np.append(np.ascontiguousarray(view).reshape(-1), L.zero)
            c = src[idx.reshape(-1)]
            o4 = off[0]
            chunks.append(c)
            off[0] += (c.size // 512) * 192
            return o4

        def addf(view, idx):                                  # -> float4 offset inside the f32 image buffer
            src = np.append(np.ascontiguousarray(view).reshape(-1), L.zero)
            c = src[idx.reshape(-1)]
            assert c.size % 4 == 0
            o4 = foff[0] // 4
            fch.append(c)
            foff[0] += c.size
            return o4

        gx = packing.gemm_index_x3
        emb_rows = packing.emb_rows_for_x3(self.E) if self.nE else 0
        d = np.zeros(DESC_INTS, np.int32)
        skip = 0
        for l, Ly in enumerate(self.layers):
            if Ly['skip']:
                skip = l
        mt = max([_tl(Ly['out']) for Ly in self.layers] + [_tl(self.Z)] + [_tl(w) for net in self.heads for w in net.widths[:2]])
        d[0:9] = [self.nE, skip, emb_rows, self.E, _tl(self.E) if self.nE else 0, mt, self.nH, _tl(self.Z), self.Z]
        d[9] = _tl(self.Z) if self.zx else 0                  # zx_tiles
        o_te, o_act, o_w, o_b, o_wb = 16, 16 + RT_MAX_L, 16 + 2 * RT_MAX_L, 16 + 3 * RT_MAX_L, 16 + 4 * RT_MAX_L
        for l, Ly in enumerate(self.layers):
            n_in = Ly['in_y'] + (self.E if Ly['skip'] else 0)
            if l == 0:
                segs = [(emb_rows, _ident(self.E))]
            elif Ly['skip']:
                segs = [(6 * _tl(Ly['in_y']), _ident(Ly['in_y'])), (emb_rows, _ident(self.E, base=Ly['in_y']))]
            else:
                segs = [(6 * _tl(Ly['in_y']), _ident(Ly['in_y']))]
            d[o_te + l], d[o_act + l] = _tl(Ly['out']), Ly['act']
            d[o_w + l] = add(L['W%d' % l].T, gx(Ly['out'], n_in, segs))
            d[o_b + l] = addf(L['b%d' % l], packing.bias_index_f16s(Ly['out']))
            if l >= 1:                                        # backward: delta_{l-1} = W_l[y part] delta_l  (rows = inputs, K = outputs)
                d[o_wb + l] = add(L['W%d' % l][:Ly['in_y'], :], gx(Ly['in_y'], Ly['out'], [(6 * _tl(Ly['out']), _ident(Ly['out']))]))
        oh = 16 + 5 * RT_MAX_L
        H = lambda field, k: oh + field * RT_MAX_H + k
        zrows = 6 * _tl(self.Z)
        for k, net in enumerate(self.heads):
            w0, w1, c = net.widths
            nout = 1 if c == 1 else 3
            W0, W1, W2 = L['H%d_W0' % k], L['H%d_W1' % k], L['H%d_W2' % k]
            pad = lambda m: np.concatenate([m, np.full((nout - c, m.shape[1]), L.zero, m.dtype)], 0) if nout > c else m
            d[H(0, k)], d[H(1, k)], d[H(2, k)] = _tl(w0), _tl(w1), c
            if self.zx:
                # kernel rows [zx (0 .. Z-1) ; z (Z .. 2Z-1)] (ref_nfr's concat order); the GEMM walks K segment X0 = z first, then X1 = zx
                Z = self.Z
                d[H(3, k)] = add(W0.T, gx(w0, 2 * Z, [(zrows, _ident(Z, base=Z)), (zrows, _ident(Z))]))
                d[10 + k] = addf(pad(W2[w1:w1 + Z].T), packing.rowdot_index_x3(nout, zrows, Z))            # offW2zx
                W0, W2 = W0[Z:], np.concatenate([W2[:w1], W2[w1 + Z:]], 0)       # from here on: the z part, as without a second input
            else:
                d[H(3, k)] = add(W0.T, gx(w0, self.Z, [(zrows, _ident(self.Z))]))
            d[H(4, k)] = add(W1.T, gx(w1, w0, [(6 * _tl(w0), _ident(w0))]))
            d[H(5, k)] = addf(L['H%d_b0' % k], packing.bias_index_f16s(w0))
            d[H(6, k)] = addf(L['H%d_b1' % k], packing.bias_index_f16s(w1))
            d[H(7, k)] = addf(pad(W2[:w1].T), packing.rowdot_index_x3(nout, 6 * _tl(w1), w1))
            d[H(8, k)] = addf(pad(W2[w1:].T), packing.rowdot_index_x3(nout, zrows, self.Z))
            d[H(9, k)] = addf(L['H%d_b2' % k], np.where(np.arange(4) < c, np.arange(4), c))
            d[H(10, k)] = add(W1, gx(w0, w1, [(6 * _tl(w1), _ident(w1))]))
            d[H(11, k)] = add(W0, gx(self.Z, w0, [(6 * _tl(w0), _ident(w0))]))
            offs = [addf(W2[:w1, j], packing.bias_index_f16s(w1)) for j in range(c)]
            d[H(12, k)] = offs[0]
            assert all(o == offs[0] + j * _tl(w1) * 8 for j, o in enumerate(offs))
            offs = [addf(W2[w1:, j], packing.bias_index_f16s(self.Z)) for j in range(c)]
            d[H(13, k)] = offs[0]
        gidx = np.concatenate(chunks)
        assert gidx.size % 512 == 0 and gidx.max() < 2 ** 31
        dev = self.device
        self._dev = (L, torch.from_numpy(gidx.astype(np.int32)).to(dev), gidx.size // 512, torch.from_numpy(np.concatenate(fch).astype(np.int32)).to(dev), d)
        return self._dev

    # ------------------------------------------------------------------ passes
    def _tensor(self, nt, tiles, dev):
        return torch.empty((nt, tiles, 32, 32), dtype=torch.float32, device=dev)

    def forward(self, x, params, zx_rows=None):
        """x: xyz [N, 3] (with an encoder) | z rows [N, Z]; zx_rows [N, Z]: the heads' second input (engines built with zx=True).
        -> state dict (saved tensors, packs), z rows | None, head outputs."""
        assert (zx_rows is not None) == self.zx, 'zx_rows exactly for an engine built with zx=True'
        L, gidx, n_steps, fidx, desc = self._static()
        N, dev = x.shape[0], x.device
        nt = (N + 31) // 32
        flat = L.flatten({n: p for n, p in zip(L.names, params)})
        pieces, wf = _C.pack_x3_gather(flat, gidx, n_steps, fidx)        # (matrices as piece triples and the thin f32 images: one launch)
        S = {'pieces': pieces, 'wf': wf, 'N': N}
        saved = []
        if self.nE:
            S['E'] = self._tensor(nt, _tl(self.E), dev)
            S['Y'] = [self._tensor(nt, _tl(Ly['out']), dev) for Ly in self.layers]
            S['ZT'] = S['Y'][-1]
            saved = [S['E']] + S['Y']
            zrows = torch.empty((N, self.Z), dtype=torch.float32, device=dev)
        else:
            S['ZT'] = self._tensor(nt, _tl(self.Z), dev)
            saved = [S['ZT']]
            zrows = None
        S['H0'] = [self._tensor(nt, _tl(net.widths[0]), dev) for net in self.heads]
        S['H1'] = [self._tensor(nt, _tl(net.widths[1]), dev) for net in self.heads]
        for a, b in zip(S['H0'], S['H1']):
            saved += [a, b]
        outs = [torch.empty((N, net.widths[2]), dtype=torch.float32, device=dev) for net in self.heads]
        S['OUT'] = outs
        S['split'] = self._split_heads(nt)
        if self.zx:
            S['ZXT'] = self._tensor(nt, _tl(self.Z), dev)
        _C.refl_train_fwd_x3(desc, pieces, wf, x if self.nE else None, None if self.nE else x, N, saved, zrows, outs, split_heads=S['split'],
                             zx_rows=zx_rows, zx_tiles_out=S.get('ZXT'))
        return S, zrows, outs

    @torch.no_grad()
    def build_packs(self, params):
        """(piece pack, f32 images) of the current weights: two launches + the flat copy; callers cache them per weights epoch."""
        L, gidx, n_steps, fidx, desc = self._static()
        flat = L.flatten({n: p.detach().float() for n, p in zip(L.names, params)})
        return _C.pack_x3_gather(flat, gidx, n_steps, fidx)

    @torch.no_grad()
    def infer(self, x, packs):
        """Inference on the same kernel (`model.matrix_mode = 'x3'`: the exact-split reflectance chain): nothing is kept for a backward.
        packs = build_packs(params).  -> (z rows | None, head outputs)."""
        L, gidx, n_steps, fidx, desc = self._static()
        x = x.detach().float().contiguous()
        N, dev = x.shape[0], x.device
        nt = (N + 31) // 32
        pieces, wf = packs
        zt = self._tensor(nt, _tl(self.Z), dev)
        if self.nE:
            saved = [None] * self.nE + [zt] + [None] * (2 * self.nH)
            zrows = torch.empty((N, self.Z), dtype=torch.float32, device=dev)
        else:
            saved = [zt] + [None] * (2 * self.nH)
            zrows = None
        outs = [torch.empty((N, net.widths[2]), dtype=torch.float32, device=dev) for net in self.heads]
        _C.refl_train_fwd_x3(desc, pieces, wf, x if self.nE else None, None if self.nE else x, N, saved, zrows, outs,
                             split_heads=self._split_heads(nt), save=False)
        return zrows, outs

    def _split_heads(self, nt):
        """Small batches: one workgroup row per head while that still fits the chip (the reference batch of 2048 points is 64 point
        tiles on 256 CUs).  VQN_REFL_SPLIT = 0 | 1 overrides."""
        import os
        force = os.environ.get('VQN_REFL_SPLIT')
        if force is not None:
            return force != '0' and self.nH > 1
        cus = _num_cus(self.device)
        return self.nH > 1 and nt * self.nH <= (3 * cus) // 2

    def backward(self, S, g_z, g_outs):
        """g_z [N, Z] | None: adjoint of z (with an encoder) / of the input rows (without: ReflStackKeepFunction) from outside the heads,
        added inside the kernel; g_outs: adjoints of the head outputs (None: zeros).
        -> (d / d input rows | None, [dW, db, dW, db, ...] in params() order, Keras layout)."""
        L, gidx, n_steps, fidx, desc = self._static()
        N, dev = S['N'], S['wf'].device
        nt = (N + 31) // 32
        g_outs = [torch.zeros_like(o) if g is None else g.detach().float().contiguous() for g, o in zip(g_outs, S['OUT'])]
        D = [self._tensor(nt, _tl(Ly['out']), dev) for Ly in self.layers]
        D0 = [self._tensor(nt, _tl(net.widths[0]), dev) for net in self.heads]
        D1 = [self._tensor(nt, _tl(net.widths[1]), dev) for net in self.heads]
        cs = [net.widths[2] for net in self.heads]
        roff = [sum(cs[:k]) for k in range(self.nH)]
        shared = self.nH > 0 and sum(cs) <= 8                  # the heads' delta_2 rows in ONE tile: z is streamed once for all of them
        if shared:
            d2all = self._tensor(nt, 1, dev)
            D2 = [d2all] * self.nH
        else:
            D2 = [self._tensor(nt, 1, dev) for net in self.heads]
        d2kw = {'d2_row0': roff} if shared else {}
        outs = list(D)
        for a, b, c in zip(D0, D1, D2):
            outs += [a, b, c]
        saved = list(S['Y']) if self.nE else []
        for a, b in zip(S['H0'], S['H1']):
            saved += [a, b]
        if g_z is not None:
            g_z = g_z.detach().float().contiguous()
        elif self.nE and not self.nH:
            g_z = torch.zeros((N, self.Z), dtype=torch.float32, device=dev)
        gz_rows = None
        args = (desc, S['pieces'], S['wf'], N, g_outs, S['OUT'])
        if S['split'] and self.nH > 1:
            # one workgroup row per head -> one d / d z slice per head; with an encoder a second launch walks it from their sum
            part = torch.empty((self.nH, N, self.Z), dtype=torch.float32, device=dev)
            _C.refl_train_bwd_x3(*args, [g_z] if (g_z is not None and not self.nE) else [], saved, outs, part, run_heads=True, run_enc=False,
                                 split_heads=True, **d2kw)
            if self.nE:
                _C.refl_train_bwd_x3(*args, [g_z] + [part[k] for k in range(self.nH)], saved, outs, None, run_heads=False, run_enc=True)
            else:
                gz_rows = part.sum(0)
        else:
            gz_rows = None if self.nE else torch.empty((N, self.Z), dtype=torch.float32, device=dev)
            _C.refl_train_bwd_x3(*args, [g_z] if g_z is not None else [], saved, outs, gz_rows, **d2kw)
        # ---- weight gradients: contractions over the points, straight into the Keras layout [in, out] ----
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        batch = WgradBatch(self.n_split, thin=True, tiles_per_block=_tiles_per_block())          # (the heads' 1..3-output last layers: vqn_wgrad_thin_batched)
        grads = []
        for l, Ly in enumerate(self.layers):
            n_in, n_out = Ly['in_y'] + (self.E if Ly['skip'] else 0), Ly['out']
            G, b = new(n_in, n_out), new(n_out)
            src = S['E'] if l == 0 else S['Y'][l - 1]
            batch.contract(D[l], src, n_out, Ly['in_y'], G, 1, n_out, bias_dst=b)
            if Ly['skip']:
                batch.contract(D[l], S['E'], n_out, self.E, G[Ly['in_y']:], 1, n_out)
            grads += [G, b]
        self._queue_heads(batch, grads, S, D0, D1, D2, shared, roff, cs, new)
        batch.flush()
        return gz_rows, grads

    def _queue_heads(self, batch, grads, S, D0, D1, D2, shared, roff, cs, new):
        z_targets, zx_targets = [], []
        for k, net in enumerate(self.heads):
            w0, w1, c = net.widths
            Z, zo = self.Z, (self.Z if self.zx else 0)            # zo: rows of the kernels in front of the z rows (the zx rows)
            g0, g1, g2, b0, b1, b2 = new(zo + Z, w0), new(w0, w1), new(w1 + zo + Z, c), new(w0), new(w1), new(c)
            batch.contract(D0[k], S['ZT'], w0, Z, g0[zo:], 1, w0, bias_dst=b0)
            if self.zx:
                batch.contract(D0[k], S['ZXT'], w0, Z, g0, 1, w0)
            batch.contract(D1[k], S['H0'][k], w1, w0, g1, 1, w1, bias_dst=b1)
            if shared:
                batch.contract_thin_rows(D2[k], roff[k], c, S['H1'][k], w1, [(0, c, g2, 1, c, b2)])
                z_targets.append((roff[k], c, g2[w1 + zo:], 1, c, None))
                if self.zx:
                    zx_targets.append((roff[k], c, g2[w1:], 1, c, None))
            else:
                batch.contract(D2[k], S['H1'][k], c, w1, g2, 1, c, bias_dst=b2)
                batch.contract(D2[k], S['ZT'], c, Z, g2[w1 + zo:], 1, c)
                if self.zx:
                    batch.contract(D2[k], S['ZXT'], c, Z, g2[w1:], 1, c)
            grads += [g0, b0, g1, b1, g2, b2]
        if shared:
            batch.contract_thin_rows(D2[0], 0, sum(cs), S['ZT'], self.Z, z_targets)
            if self.zx:
                batch.contract_thin_rows(D2[0], 0, sum(cs), S['ZXT'], self.Z, zx_targets)


class ReflStackFunction(torch.autograd.Function):
    """(engine, x, *engine.params()) -> (z rows, head outputs ...) with an encoder, (head outputs ...) without."""

    @staticmethod
    def forward(ctx, engine, x, *params):
        with torch.no_grad():
            S, zrows, outs = engine.forward(x.detach().float().contiguous(), [p.detach().float() for p in params])
        ctx.engine, ctx.S = engine, S
        ctx.x_needs = x.requires_grad and not engine.nE
        return ((zrows,) if engine.nE else ()) + tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        eng = ctx.engine
        g_z, g_outs = (gs[0], gs[1:]) if eng.nE else (None, gs)
        with torch.no_grad():
            gz_rows, grads = eng.backward(ctx.S, g_z, list(g_outs))
        ctx.S = None
        return (None, gz_rows if not eng.nE else None) + tuple(grads)


class ReflStackZxFunction(torch.autograd.Function):
    """Stage 3: (engine, x, zx, *engine.params()) -> (z rows, head outputs ...): encoder on x (the reference colours through rgb_enc) -> z, heads
    over [zx ; z].  zx (z_xyz of the frozen stage-2 encoder) gets NO adjoint: callers pass rows that do not require one."""

    @staticmethod
    def forward(ctx, engine, x, zx, *params):
        assert engine.zx and engine.nE and not zx.requires_grad
        with torch.no_grad():
            S, zrows, outs = engine.forward(x.detach().float().contiguous(), [p.detach().float() for p in params],
                                            zx_rows=zx.detach().float().contiguous())
        ctx.engine, ctx.S = engine, S
        return (zrows,) + tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        eng = ctx.engine
        with torch.no_grad():
            _, grads = eng.backward(ctx.S, gs[0], list(gs[1:]))
        ctx.S = None
        return (None, None, None) + tuple(grads)


class ReflStackKeepFunction(torch.autograd.Function):
    """Heads-only stack that also hands its INPUT rows on: (engine, x, *params) -> (x, head outputs ...).  A second consumer of the rows (the
    smoothness term of the loss reads the quantised z the VQ heads read) then hangs off this node instead of off the rows' producer, its
    adjoint arrives here, and the backward kernel adds it to d / d rows on the fly (`g_z_rows` of vqn_refl_train_bwd_x3) -- the autograd
    engine's own accumulation was an [N, 256] framework addition per step (67 M elements at the 262,144-point batch)."""

    @staticmethod
    def forward(ctx, engine, x, *params):
        assert not engine.nE
        with torch.no_grad():
            S, _, outs = engine.forward(x.detach().float().contiguous(), [p.detach().float() for p in params])
        ctx.engine, ctx.S = engine, S
        return (x,) + tuple(outs)

    @staticmethod
    def backward(ctx, g_keep, *g_outs):
        eng = ctx.engine
        with torch.no_grad():
            gz_rows, grads = eng.backward(ctx.S, g_keep, list(g_outs))
        ctx.S = None
        return (None, gz_rows) + tuple(grads)
