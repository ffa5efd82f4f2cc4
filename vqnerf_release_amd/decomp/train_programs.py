"""Training passes of the reflectance Dense stacks as tile programs (csrc/tile_vm.hip + csrc/wgrad.hip): what the
reference gets from `tape.gradient` through networks/mlp.py:24-50 as used by vq_nfr.py:771-828 (train_nfr.py:562-576).

Two engines, each with a forward program that saves every activation in TFMT and a reverse-sweep program:
  EncoderEngine : xyz -> posenc -> fine_enc (skip-concat) -> bottleneck -> z            (weights' gradients only; xyz is data)
  HeadsEngine   : z -> {diff, spec, rough} heads (skip-concat of z into the last layer) (weights' gradients and d/dz)
Weights keep the Keras layout (kernel [in, out]); sigmoid' / relu' of the LAST layer of a stack is applied by the caller
in torch before the reverse program (one elementwise op on [N, <=256])."""
import ctypes

import numpy as np
import torch

from vqnerf_release_amd import _C
from vqnerf_release_amd.geo.train_programs import (Program, _ident, _f2i, DESC_INTS, K_LD_POSENC, K_LD_T, EPI_ACT, EPI_MUL_DACT,
                                                   ACT_NONE, ACT_RELU, ACT_SIGMOID, FlatLayout, build_static_packs, WGRAD_ENTRY, wgrad_mode,
                                                   WgradBatch, BATCHED_WGRAD)

ACTS = {None: ACT_NONE, 'relu': ACT_RELU, 'sigmoid': ACT_SIGMOID}


def to_tfmt(x, tiles=None, out=None):
    """[N, F] row-major -> TFMT [ceil(N/32), ceil(F/32), 32 features, 32 points] (zero padded): one transpose kernel."""
    N, F = x.shape
    nt, ft = (N + 31) // 32, tiles or (F + 31) // 32
    x = x.detach()
    if x.dtype != torch.float32 or x.stride(1) != 1:
        x = x.float().contiguous()
    if out is None:
        out = torch.empty((nt, ft, 32, 32), dtype=torch.float32, device=x.device)
    assert tuple(out.shape) == (nt, ft, 32, 32) and out.is_contiguous()
    _C.require_device(x, 'to_tfmt')
    rc = _C.lib().vqn_tfmt_pack(_C._ptr(x), ctypes.c_int64(N), ctypes.c_int(F), ctypes.c_int64(x.stride(0) if N > 1 else F),
                                _C._ptr(out), ctypes.c_int(ft), _C._stream())
    _C._check(rc, 'vqn_tfmt_pack')
    return out


def pack_delta(g, y_tfmt, act, N, F, out):
    """out (TFMT) = g act'(y): the adjoint rows g [N, F] (None: zeros) times the activation derivative taken from the layer's saved
    TFMT output, in one launch (vqn_tfmt_pack_delta)."""
    if g is not None:
        g = g.detach()
        if g.dtype != torch.float32 or g.stride(1) != 1:
            g = g.float().contiguous()
        _C.require_device(g, 'pack_delta')
    assert out.is_contiguous() and y_tfmt.is_contiguous() and out.shape == y_tfmt.shape
    with _C._clock('vqn_tfmt_pack_delta'):
        rc = _C.lib().vqn_tfmt_pack_delta(_C._ptr(g), ctypes.c_int64(N), ctypes.c_int(F), ctypes.c_int64((g.stride(0) if N > 1 else F) if g is not None else F),
                                          _C._ptr(y_tfmt), ctypes.c_int(act), _C._ptr(out), ctypes.c_int(out.shape[1]), _C._stream())
    _C._check(rc, 'vqn_tfmt_pack_delta')
    return out


def from_tfmt(t, N, F):
    """TFMT -> contiguous [N, F] rows."""
    nt, ft = t.shape[0], t.shape[1]
    assert t.is_contiguous() and nt * 32 >= N and ft * 32 >= F
    out = torch.empty((N, F), dtype=torch.float32, device=t.device)
    rc = _C.lib().vqn_tfmt_unpack(_C._ptr(t), ctypes.c_int(ft), ctypes.c_int64(N), ctypes.c_int(F), _C._ptr(out), ctypes.c_int64(F),
                                  _C._stream())
    _C._check(rc, 'vqn_tfmt_unpack')
    return out


class _Engine:
    """shared: program packing / launching / weight-gradient contraction"""
    n_split = 256        # upper bound on the split-over-points workgroups of a weight-gradient call (one per CU); a call uses
                         # min(n_split, point tiles), so the reference batch (2048 points = 64 tiles) still runs 64 of them

    def _finish(self, progs):
        for name, build in progs:
            prog = build()
            assert prog.total_rows * 1024 <= 160 * 1024, f'{name}: {prog.total_rows} KB of LDS'
            setattr(self, name, prog.materialize())
        self._dev = {}

    def _static(self, names):
        """weight-independent part of the packs (global gather index + descriptors), built once."""
        if 'static' not in self._dev:
            L = self.layout()
            progs = {n: getattr(self, n) for n in names}
            gidx, descs = build_static_packs(progs, L, lambda key: self.mat_index(key, L))
            self._dev['static'] = (L, torch.from_numpy(gidx).to(self.device),
                                   {n: (d, torch.from_numpy(d).to(self.device)) for n, d in descs.items()})
        return self._dev['static']

    def pack(self, names, sources):
        """sources: dict name -> tensor as declared by layout().  ONE gather builds the whole weight buffer."""
        L, gidx, descs = self._static(names)
        return L.flatten(sources)[gidx], descs

    def run(self, which, descs, wbuf, tensors, specs, N):
        prog = getattr(self, which)
        d_host, d_dev = descs[which]
        names = list(prog.tn.keys())
        ptrs = (ctypes.c_void_p * len(names))(*[tensors[n].data_ptr() if n in tensors else 0 for n in names])
        lds = np.array([specs[n][1] for n in names], np.int32)
        with _C._clock('vqn_tile_program:' + type(self).__name__ + '.' + which):
            rc = _C.lib().vqn_tile_program(ctypes.c_void_p(d_dev.data_ptr()), d_host.ctypes.data_as(ctypes.c_void_p), _C._ptr(wbuf),
                                           ptrs, lds.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(len(names)), ctypes.c_int64(N),
                                           _C._stream())
        _C._check(rc, 'vqn_tile_program')

    def wgrad(self, A, B, a_rows, b_cols, ws, rowsum=False):
        """sum_p A[o][p] B[i][p] -> [a_rows, b_cols]; rowsum=True: also sum_p A[o][p] (bias gradient) from the same pass."""
        nt, at, bt = A.shape[0], A.shape[1], B.shape[1]
        an, bn = (a_rows + 31) // 32, (b_cols + 31) // 32
        assert an <= 8 and bn <= 8
        if rowsum and (getattr(self, '_rs_ws', None) is None or self._rs_ws.device != A.device):
            self._rs_ws = torch.empty(self.n_split * 256, dtype=torch.float32, device=A.device)
        lib = _C.lib()
        with _C._clock(WGRAD_ENTRY[wgrad_mode()]):
            n = getattr(lib, WGRAD_ENTRY[wgrad_mode()])(_C._ptr(A), ctypes.c_int(at), ctypes.c_int(0), ctypes.c_int(an), _C._ptr(B), ctypes.c_int(bt),
                                       ctypes.c_int(0), ctypes.c_int(bn), ctypes.c_int64(nt), ctypes.c_int(self.n_split),
                                       _C._ptr(ws), _C._ptr(self._rs_ws if rowsum else None), _C._stream())
        if n <= 0:
            _C._check(n if n < 0 else -3, 'vqn_wgrad_partials')
        out = torch.empty((an * 32, bn * 32), dtype=torch.float32, device=A.device)
        with _C._clock('vqn_reduce_partials'):                        # ordered sum of the split-over-points partial blocks
            rc = lib.vqn_reduce_partials(_C._ptr(ws), ctypes.c_int(n), ctypes.c_int(an * 32), ctypes.c_int(bn * 32), _C._ptr(out),
                                         ctypes.c_int64(bn * 32), ctypes.c_int(0), _C._stream())
            _C._check(rc, 'vqn_reduce_partials')
            if rowsum:
                rs = torch.empty((an * 32,), dtype=torch.float32, device=A.device)
                rc = lib.vqn_reduce_partials(_C._ptr(self._rs_ws), ctypes.c_int(n), ctypes.c_int(1), ctypes.c_int(an * 32), _C._ptr(rs),
                                             ctypes.c_int64(an * 32), ctypes.c_int(0), _C._stream())
                _C._check(rc, 'vqn_reduce_partials')
                return out[:a_rows, :b_cols], rs[:a_rows]
        return out[:a_rows, :b_cols]

    @staticmethod
    def alloc(specs, N, device, only=None):
        nt = (N + 31) // 32
        out = {}
        for name, (kind, w) in specs.items():
            if only is not None and name not in only:
                continue
            out[name] = torch.empty((N, w), dtype=torch.float32, device=device) if kind == 'vec' \
                else torch.empty((nt, w, 32, 32), dtype=torch.float32, device=device)
        return out


class EncoderEngine(_Engine):
    def __init__(self, fine_enc, bottleneck, n_freqs, device):
        self.device = device
        self.nets = [fine_enc, bottleneck]
        self.E = 3 + 6 * n_freqs
        self.mr = n_freqs
        # flat layer list: (net idx, layer idx, in_feats of the y-part, out, act, takes_skip_input)
        self.layers = []
        d_prev = self.E
        for ni, net in enumerate(self.nets):
            d_in_net = d_prev
            for li, (w, a) in enumerate(zip(net.widths, net.act)):
                skip_in = net.skip_at is not None and (li - 1) in net.skip_at      # input of layer li = [y_{li-1} ; net input]
                assert not (skip_in and ni != 0), 'skip-concat of a non-posenc input is not needed by the shipped nets'
                self.layers.append(dict(net=ni, li=li, in_y=d_prev, out=w, act=ACTS[a], skip=skip_in))
                d_prev = w
            assert d_in_net is not None
        self.nl = len(self.layers)
        self.specs = {'X': ('vec', 3), 'E': ('t', (self.E + 31) // 32), 'GZ': ('t', (self.layers[-1]['out'] + 31) // 32)}
        for k, L in enumerate(self.layers):
            self.specs['Y%d' % k] = ('t', (L['out'] + 31) // 32)
            self.specs['D%d' % k] = ('t', (L['out'] + 31) // 32)
        self._finish([('prog_fwd', self._build_fwd), ('prog_bwd', self._build_bwd)])

    def _build_fwd(self):
        P = Program(list(self.specs.keys()))
        rE = P.alloc(self.E, [])
        P.op(K_LD_POSENC, P.t('X'), P.row(rE), self.mr, self.E, P.t('E'), _f2i(1.0))
        prev = rE
        for k, L in enumerate(self.layers):
            if L['skip']:
                segs, cols, shape = [prev, rE], [_ident(L['in_y']), _ident(self.E, base=L['in_y'])], (L['out'], L['in_y'] + self.E)
            else:
                segs, cols, shape = [prev], [_ident(L['in_y'])], (L['out'], L['in_y'])
            prev = P.gemm(('Wt', k), shape, segs, cols, L['out'], live=[rE], act=L['act'], bias_key=('b', k), store='Y%d' % k)
        return P.finalize()

    def _build_bwd(self):
        P = Program(list(self.specs.keys()))
        top = self.nl - 1
        r = P.alloc(self.layers[top]['out'], [], tiles=(self.layers[top]['out'] + 31) // 32)
        P.op(K_LD_T, P.t('GZ'), P.row(r), r.rows)                       # = delta_top (act' of the top layer applied by the caller)
        for k in range(top, 0, -1):                                   # delta_{k-1} = (W_k[y-part] delta_k) * act'_{k-1}(Y_{k-1})
            L, Lp = self.layers[k], self.layers[k - 1]
            r = P.gemm(('Wy', k), (L['in_y'], L['out']), [r], [_ident(L['out'])], L['in_y'], live=[], epi=EPI_MUL_DACT, act=Lp['act'],
                       aux1='Y%d' % (k - 1), store='D%d' % (k - 1))
        return P.finalize()

    def _kernels(self):
        ks = []
        for net in self.nets:
            for layer in net.layers:
                ks.append((layer.kernel, layer.bias))
        return ks

    def layout(self):
        shp = []
        for k, L in enumerate(self.layers):
            shp.append(('W%d' % k, (L['in_y'] + (self.E if L['skip'] else 0), L['out'])))
        for k, L in enumerate(self.layers):
            shp.append(('b%d' % k, (L['out'],)))
        return FlatLayout(shp)

    def mat_index(self, key, L):
        kind, k = key
        if kind == 'Wt':
            return L['W%d' % k].T
        if kind == 'b':
            return L['b%d' % k]
        if kind == 'Wy':
            return L['W%d' % k][:self.layers[k]['in_y'], :]
        raise KeyError(key)

    def forward(self, xyz, W, b):
        N = xyz.shape[0]
        src = {'W%d' % k: w for k, w in enumerate(W)}
        src.update({'b%d' % k: t for k, t in enumerate(b)})
        wbuf, descs = self.pack(['prog_fwd', 'prog_bwd'], src)
        T = self.alloc(self.specs, N, xyz.device)
        T['X'].copy_(xyz)
        self.run('prog_fwd', descs, wbuf, T, self.specs, N)
        return T, wbuf, descs

    def backward(self, T, wbuf, descs, g_z, N):
        top = self.nl - 1
        pack_delta(g_z, T['Y%d' % top], self.layers[top]['act'], N, self.layers[top]['out'], T['GZ'])
        T['D%d' % top] = T['GZ']
        self.run('prog_bwd', descs, wbuf, T, self.specs, N)
        if not BATCHED_WGRAD[0]:
            ws = torch.empty(min(self.n_split, (N + 31) // 32) * 256 * 256, dtype=torch.float32, device=g_z.device)
        dW, db, dev = [], [], g_z.device
        batch = WgradBatch(self.n_split)
        for k, L in enumerate(self.layers):
            D = T['D%d' % k]
            src = T['E'] if k == 0 else T['Y%d' % (k - 1)]
            if BATCHED_WGRAD[0]:
                # straight into the Keras layout [in, out] (element (o, i) at i * out + o), the skip input's rows after the y-part's
                n_in, n_out = L['in_y'] + (self.E if L['skip'] else 0), L['out']
                G = torch.empty((n_in, n_out), dtype=torch.float32, device=dev)
                bsum = torch.empty((n_out,), dtype=torch.float32, device=dev)
                batch.contract(D, src, n_out, L['in_y'], G, 1, n_out, bias_dst=bsum)
                if L['skip']:
                    batch.contract(D, T['E'], n_out, self.E, G[L['in_y']:], 1, n_out)
                dW.append(G)
                db.append(bsum)
                continue
            g, bsum = self.wgrad(D, src, L['out'], L['in_y'], ws, rowsum=True)
            if L['skip']:
                g = torch.cat([g, self.wgrad(D, T['E'], L['out'], self.E, ws)], 1)
            dW.append(g.t().contiguous())                              # Keras layout [in, out]
            db.append(bsum)
        batch.flush()
        return dW, db


class EncoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, xyz, *params):
        n = engine.nl
        W, b = [p.detach().float() for p in params[:n]], [p.detach().float() for p in params[n:]]
        with torch.no_grad():
            T, wbuf, descs = engine.forward(xyz.detach().float().contiguous(), W, b)
        ctx.engine, ctx.T, ctx.wbuf, ctx.descs, ctx.N = engine, T, wbuf, descs, xyz.shape[0]
        top = n - 1
        return from_tfmt(T['Y%d' % top], xyz.shape[0], engine.layers[top]['out']).contiguous()

    @staticmethod
    def backward(ctx, g_z):
        with torch.no_grad():
            dW, db = ctx.engine.backward(ctx.T, ctx.wbuf, ctx.descs, g_z.contiguous(), ctx.N)
        ctx.T = None
        return (None, None) + tuple(dW) + tuple(db)


class HeadsEngine(_Engine):
    """Heads of one family sharing the input z: widths [z, z/2, c], relu relu sigmoid, skip_at=[1] (input of the last
    layer = [y1 ; z])."""

    def __init__(self, nets, z_dim, device):
        self.device, self.nets, self.Z = device, nets, z_dim
        for net in nets:
            assert len(net.widths) == 3 and net.skip_at == [1] and net.act == ['relu', 'relu', 'sigmoid']
        self.specs = {'Z': ('t', (z_dim + 31) // 32), 'GZ': ('t', (z_dim + 31) // 32)}
        for h, net in enumerate(nets):
            for k, w in enumerate(net.widths):
                self.specs['Y%d_%d' % (h, k)] = ('t', (w + 31) // 32)
                self.specs['D%d_%d' % (h, k)] = ('t', (w + 31) // 32)
        self._finish([('prog_fwd', self._build_fwd), ('prog_bwd', self._build_bwd)])

    def _build_fwd(self):
        P = Program(list(self.specs.keys()))
        rZ = P.alloc(self.Z, [])
        P.op(K_LD_T, P.t('Z'), P.row(rZ), rZ.rows)
        for h, net in enumerate(self.nets):
            w0, w1, c = net.widths
            y0 = P.gemm(('Wt', h, 0), (w0, self.Z), [rZ], [_ident(self.Z)], w0, live=[rZ], act=ACT_RELU, bias_key=('b', h, 0), store='Y%d_0' % h)
            y1 = P.gemm(('Wt', h, 1), (w1, w0), [y0], [_ident(w0)], w1, live=[rZ], act=ACT_RELU, bias_key=('b', h, 1), store='Y%d_1' % h)
            P.gemm(('Wt', h, 2), (c, w1 + self.Z), [y1, rZ], [_ident(w1), _ident(self.Z, base=w1)], c, live=[rZ], act=ACT_SIGMOID,
                   bias_key=('b', h, 2), store='Y%d_2' % h, want_dst=False)
        return P.finalize()

    def _build_bwd(self):
        P = Program(list(self.specs.keys()))
        rGZ = None
        last = len(self.nets) - 1
        for h, net in enumerate(self.nets):
            w0, w1, c = net.widths
            r2 = P.alloc(c, [rGZ] if rGZ else [], tiles=1)
            P.op(K_LD_T, P.t('D%d_2' % h), P.row(r2), r2.rows)           # delta_2 = g_out * sigmoid'(out), from the caller
            keep = [rGZ] if rGZ else []
            # z-part of the last layer's input: GZ (+)= W2[w1:, :] delta_2
            if rGZ is None:
                rGZ = P.gemm(('W2z', h), (self.Z, c), [r2], [_ident(c)], self.Z, live=[r2])
            else:
                P.gemm(('W2z', h), (self.Z, c), [r2], [_ident(c)], self.Z, live=[r2], dst=rGZ, accumulate=True)
            d1 = P.gemm(('W2y', h), (w1, c), [r2], [_ident(c)], w1, live=[rGZ], epi=EPI_MUL_DACT, act=ACT_RELU, aux1='Y%d_1' % h,
                        store='D%d_1' % h)
            d0 = P.gemm(('W1', h), (w0, w1), [d1], [_ident(w1)], w0, live=[rGZ], epi=EPI_MUL_DACT, act=ACT_RELU, aux1='Y%d_0' % h,
                        store='D%d_0' % h)
            P.gemm(('W0', h), (self.Z, w0), [d0], [_ident(w0)], self.Z, live=[], dst=rGZ, accumulate=True,
                   store='GZ' if h == last else None)
        return P.finalize()

    def layout(self):
        shp = []
        for h, net in enumerate(self.nets):
            w0, w1, c = net.widths
            shp += [('W%d_0' % h, (self.Z, w0)), ('W%d_1' % h, (w0, w1)), ('W%d_2' % h, (w1 + self.Z, c)),
                    ('b%d_0' % h, (w0,)), ('b%d_1' % h, (w1,)), ('b%d_2' % h, (c,))]
        return FlatLayout(shp)

    def mat_index(self, key, L):
        kind, h = key[0], key[1]
        w1 = self.nets[h].widths[1]
        if kind == 'Wt':
            return L['W%d_%d' % (h, key[2])].T
        if kind == 'b':
            return L['b%d_%d' % (h, key[2])]
        if kind == 'W2z':
            return L['W%d_2' % h][w1:, :]
        if kind == 'W2y':
            return L['W%d_2' % h][:w1, :]
        if kind == 'W1':
            return L['W%d_1' % h]
        if kind == 'W0':
            return L['W%d_0' % h]
        raise KeyError(key)

    def forward(self, z, W, b):
        N = z.shape[0]
        src = {}
        for h in range(len(self.nets)):
            for k in range(3):
                src['W%d_%d' % (h, k)], src['b%d_%d' % (h, k)] = W[h][k], b[h][k]
        wbuf, descs = self.pack(['prog_fwd', 'prog_bwd'], src)
        T = self.alloc(self.specs, N, z.device)
        to_tfmt(z, self.specs['Z'][1], out=T['Z'])
        self.run('prog_fwd', descs, wbuf, T, self.specs, N)
        return T, wbuf, descs

    def backward(self, T, wbuf, descs, g_outs, N):
        for h, (net, g) in enumerate(zip(self.nets, g_outs)):
            pack_delta(g, T['Y%d_2' % h], ACT_SIGMOID, N, net.widths[2], T['D%d_2' % h])
        self.run('prog_bwd', descs, wbuf, T, self.specs, N)
        if not BATCHED_WGRAD[0]:
            ws = torch.empty(min(self.n_split, (N + 31) // 32) * 256 * 256, dtype=torch.float32, device=T['Z'].device)
        dW, db, dev = [], [], T['Z'].device
        batch = WgradBatch(self.n_split)
        for h, net in enumerate(self.nets):
            w0, w1, c = net.widths
            D0, D1, D2 = T['D%d_0' % h], T['D%d_1' % h], T['D%d_2' % h]
            if BATCHED_WGRAD[0]:
                new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
                g0, g1, g2, b0, b1, b2 = new(self.Z, w0), new(w0, w1), new(w1 + self.Z, c), new(w0), new(w1), new(c)
                batch.contract(D0, T['Z'], w0, self.Z, g0, 1, w0, bias_dst=b0)
                batch.contract(D1, T['Y%d_0' % h], w1, w0, g1, 1, w1, bias_dst=b1)
                batch.contract(D2, T['Y%d_1' % h], c, w1, g2, 1, c, bias_dst=b2)
                batch.contract(D2, T['Z'], c, self.Z, g2[w1:], 1, c)
                dW += [g0, g1, g2]
                db += [b0, b1, b2]
                continue
            g0, b0 = self.wgrad(D0, T['Z'], w0, self.Z, ws, rowsum=True)
            g1, b1 = self.wgrad(D1, T['Y%d_0' % h], w1, w0, ws, rowsum=True)
            g2a, b2 = self.wgrad(D2, T['Y%d_1' % h], c, w1, ws, rowsum=True)
            g0, g1 = g0.t().contiguous(), g1.t().contiguous()
            g2 = torch.cat([g2a, self.wgrad(D2, T['Z'], c, self.Z, ws)], 1).t().contiguous()
            dW += [g0, g1, g2]
            db += [b0, b1, b2]
        batch.flush()
        return from_tfmt(T['GZ'], N, self.Z), dW, db


class HeadsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, z, *params):
        nh = len(engine.nets)
        W = [[p.detach().float() for p in params[3 * h:3 * h + 3]] for h in range(nh)]
        b = [[p.detach().float() for p in params[3 * nh + 3 * h:3 * nh + 3 * h + 3]] for h in range(nh)]
        with torch.no_grad():
            T, wbuf, descs = engine.forward(z.detach().float().contiguous(), W, b)
        ctx.engine, ctx.T, ctx.wbuf, ctx.descs, ctx.N = engine, T, wbuf, descs, z.shape[0]
        return tuple(from_tfmt(T['Y%d_2' % h], z.shape[0], net.widths[2]).contiguous() for h, net in enumerate(engine.nets))

    @staticmethod
    def backward(ctx, *g_outs):
        with torch.no_grad():
            g_z, dW, db = ctx.engine.backward(ctx.T, ctx.wbuf, ctx.descs, g_outs, ctx.N)
        ctx.T = None
        return (None, g_z.contiguous()) + tuple(dW) + tuple(db)
