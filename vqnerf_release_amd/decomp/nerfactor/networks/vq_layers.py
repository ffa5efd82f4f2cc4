"""VectorQuantizerEMA on MI355X: mirror of
decomp/nerfvq_nfr3/nerfactor/networks/vq_layers.py:174-349 (same constructor arguments, same call
signature and result keys), with the nearest-code search and the EMA statistics running in the HIP
kernels `vqn_vq_assign` / `vqn_vq_ema_stats` (csrc/vq.hip).

Differences that are deliberate and documented:
  * the code-dropout draw ``tf.random.uniform((1, K))`` (vq_layers.py:287) is an explicit optional
    argument ``roll`` so that runs are reproducible and data-parallel ranks can share it;
  * no host synchronisation: the reference's ``.numpy()`` at vq_layers.py:318 is gone;
  * ``ExponentialMovingAverage`` restates dm-sonnet 2.0.0's ``sonnet.src.moving_averages`` (third
    party, not in the reference tree): hidden -= (hidden - v) * (1 - decay); counter += 1;
    average = hidden / (1 - decay ** counter).
"""
import torch

from vqnerf_release_amd import _C


class ExponentialMovingAverage(torch.nn.Module):
    def __init__(self, decay, shape, dtype=torch.float32):
        super().__init__()
        self.decay = float(decay)
        self.register_buffer('hidden', torch.zeros(shape, dtype=dtype))
        self.register_buffer('average', torch.zeros(shape, dtype=dtype))
        self.register_buffer('counter', torch.zeros((), dtype=torch.int64))

    def initialize(self, value):
        self.hidden = torch.zeros_like(value)
        self.average = torch.zeros_like(value)

    @torch.no_grad()
    def update(self, value):
        """hidden -= (hidden - v) (1 - decay); counter += 1; average = hidden / (1 - decay^counter).  Everything stays on the
        device (no host copy of the counter: the update can live inside a captured HIP graph); the zero-debias factor is taken
        in float64 so that 1 - 0.999^k keeps its digits."""
        self.counter += 1
        self.hidden -= (self.hidden - value) * (1.0 - self.decay)
        debias = 1.0 - torch.pow(torch.full((), self.decay, dtype=torch.float64, device=self.counter.device), self.counter.double())
        self.average = (self.hidden.double() / debias).to(self.hidden.dtype)

    @property
    def value(self):
        return self.average

    def forward(self, value):
        self.update(value)
        return self.average


class _SteAndCommitment(torch.autograd.Function):
    """(inputs + sg(q - inputs), mean((sg(q) - inputs)^2)) in one fused pass (vq_layers.py:302, :327); gradients: identity
    through the first output, 2 (inputs - q) / numel through the second."""

    @staticmethod
    def forward(ctx, inputs, quant):
        x = inputs.detach().contiguous()
        ste, loss = _C.vq_ste_loss(x.reshape(quant.shape), quant)
        ctx.save_for_backward(x, quant)
        return ste.reshape(inputs.shape), loss

    @staticmethod
    def backward(ctx, g_ste, g_loss):
        x, q = ctx.saved_tensors
        g = g_ste
        if g_loss is not None and x.is_cuda and x.numel() % 4 == 0 and g_loss.dim() == 0:
            gs = None if g_ste is None else g_ste.reshape(x.shape).float().contiguous()
            return _C.vq_ste_loss_bwd(x, q.reshape(x.shape).contiguous(), gs, g_loss.float().contiguous()), None
        if g_loss is not None:
            gl = (x - q.reshape(x.shape)) * (g_loss * (2.0 / x.numel()))
            g = gl if g is None else g + gl
        return g, None


class _QuantiseTrain(torch.autograd.Function):
    """The training path's normalise -> nearest code -> straight-through + commitment chain (vq_nfr.py:575-578, vq_layers.py:277-302,
    :327) as ONE pass over the rows (`vqn_vq_quantize_rows_train`; was l2-normalise, assign, straight-through, loss-final: four launches
    and 7 KB per row) -- bit-identical indices, straight-through rows and loss to that sequence (tests/test_gpu_vq.py).  Outputs:
    straight-through rows, mean((sg(q) - x^)^2); `aux` receives idx / counts / the normalised rows (not differentiable).
    Backward = the two kernels of the sequence (`vqn_vq_ste_loss_bwd`, `vqn_l2_normalize_rows_bwd`) as one pass (`vqn_vq_train_bwd`); it reads q back as the saved
    straight-through rows x^ + (q - x^), equal to q to an ulp (the commitment gradient 2 beta (x^ - q) / numel sees a 1e-7 relative
    difference)."""

    @staticmethod
    def forward(ctx, z_raw, cb, sel, eps, aux, loss_post=1.0):
        """loss_post: the commitment cost -- the returned loss is commitment_cost * mean(...), rounded as the framework's own
        scalar multiplication would round it (that multiplication and its adjoint were two launches on one number)."""
        x = z_raw.detach().float().contiguous()
        idx, ste, loss, counts, xnorm = _C.vq_quantize_rows(x, cb, sel_mask=sel, eps=eps, want_ste=True, want_xnorm=True, loss_post=loss_post)
        aux.update(idx=idx, counts=counts, xnorm=xnorm)
        ctx.save_for_backward(x, xnorm, ste)
        ctx.eps, ctx.post = float(eps), float(loss_post)
        return ste, loss

    @staticmethod
    def backward(ctx, g_ste, g_loss):
        x, xnorm, ste = ctx.saved_tensors
        gs = None if g_ste is None else g_ste.float().contiguous()
        gl = torch.zeros((), dtype=torch.float32, device=x.device) if g_loss is None else g_loss.float().contiguous()
        if x.shape[1] <= 1024:
            return _C.vq_train_bwd(x, xnorm, ste, gs, gl, ctx.eps, loss_post=ctx.post), None, None, None, None, None     # (one pass: the two kernels below)
        g_xn = _C.vq_ste_loss_bwd(xnorm, ste, gs, gl if ctx.post == 1.0 else gl * ctx.post)
        return _C.l2_normalize_rows_bwd(x, g_xn, ctx.eps), None, None, None, None, None


class L2NormalizeRows(torch.autograd.Function):
    """`safe_l2_normalize(x, axis=1)` (util/math.py:63-64) on `vqn_l2_normalize_rows`: the sum of squares in the DEFINED order the
    VQ kernels use for |x|^2, so that the fused inference kernel (`vqn_vq_quantize_rows`) and this three-kernel sequence agree bit
    for bit.  Backward: y = x s, s = max(sum x^2, eps)^(-1/2)  ->  g s - x s^3 (x . g) where sum x^2 > eps, g s elsewhere."""

    @staticmethod
    def forward(ctx, x, eps):
        xd = x.detach().float().contiguous()
        ctx.save_for_backward(xd)
        ctx.eps = float(eps)
        return _C.l2_normalize_rows(xd, ctx.eps)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if x.is_cuda and x.shape[1] % 4 == 0 and x.shape[1] <= 1024:
            return _C.l2_normalize_rows_bwd(x, g.float().contiguous(), ctx.eps), None
        x2 = (x * x).sum(1, keepdim=True)
        s = torch.rsqrt(torch.clamp(x2, min=ctx.eps))
        gx = g * s - torch.where(x2 > ctx.eps, x * (s * s * s) * (x * g).sum(1, keepdim=True), torch.zeros_like(x))
        return gx, None


def l2_normalize_rows(x, eps=1e-6):
    """Row-wise l2-normalisation of a [N, D] device tensor on the HIP kernel (D % 4 == 0), differentiable."""
    return L2NormalizeRows.apply(x, eps)


class LazyResult(dict):
    """The result dict of the VQ layer with some values computed on first access: the model only ever reads `quantize`, `loss`,
    `encoding_indices` (and `update` in training); `encodings` (an [N, K] one-hot), `distances` ([N, K]) and `perplexity` cost
    passes over the rows that nobody pays for unless they look."""

    def __init__(self, values, thunks):
        super().__init__(values)
        self._thunks = dict(thunks)

    def __missing__(self, key):
        if key in self._thunks:
            self[key] = self._thunks.pop(key)()
            return dict.__getitem__(self, key)
        raise KeyError(key)

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._thunks

    def keys(self):
        return list(dict.keys(self)) + list(self._thunks)


class LazyKwargs(LazyResult):
    """A keyword dict with values that exist only if somebody indexes them: `d['z']` computes, `f(**d)` / iteration / `keys()` see the
    materialised entries only (a consumer that never names the key never pays for it)."""

    def keys(self):
        return list(dict.keys(self))

    def pop(self, key, *default):
        if not dict.__contains__(self, key) and key in self._thunks:
            return self._thunks.pop(key)()
        return dict.pop(self, key, *default)


class VectorQuantizerEMA(torch.nn.Module):
    def __init__(self, embedding_dim, num_embeddings, commitment_cost, seed, decay=0.999, epsilon=1e-5,
                 dtype=torch.float32, name='vector_quantizer_ema'):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.num_embeddings = num_embeddings
        if not 0 <= decay <= 1:
            raise ValueError('decay must be in range [0, 1]')
        self.decay = decay
        self.commitment_cost = commitment_cost
        self.epsilon = epsilon
        self.name = name
        self._gen = torch.Generator(device='cpu')
        self._gen.manual_seed(int(seed))
        self.ema_cluster_size = ExponentialMovingAverage(decay, (num_embeddings,), dtype)
        self.ema_dw = ExponentialMovingAverage(decay, (embedding_dim, num_embeddings), dtype)
        # hook for data-parallel training: called with the flat [K + D*K] statistics buffer
        # (sum over ranks) before the EMAs are updated -- see parallel.py
        self.stats_all_reduce = None
        self.fuse_ema_update = True      # training on the device: the EMA codebook move as one launch (False: the torch statement)

    def draw_roll(self, individual=True, device=None):
        """The code-dropout draw tf.random.uniform((1, K)) (vq_layers.py:287).  Normally from the layer's seeded host generator;
        inside a HIP-graph capture from the device generator (a host draw + copy cannot be captured; torch advances the
        philox offset of captured draws at every replay)."""
        n = self.num_embeddings if individual else 1
        dev = torch.device(device or 'cpu')
        if dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            return torch.rand((1, n), device=dev)
        return torch.rand((1, n), generator=self._gen).to(dev)

    def forward(self, inputs, codebook, is_training, thres=None, individual=True, roll=None,
                return_distances=True, raw=False):
        """`raw=True` (training only): `inputs` are the UN-normalised encoder rows and the layer normalises them itself, in the same
        pass as the nearest-code search (`train_from_raw`); going through `forward` keeps module hooks working."""
        if raw:
            assert is_training, 'raw=True is the training form (inference: infer_from_raw)'
            return self.train_from_raw(inputs, codebook, thres=thres, individual=individual, roll=roll)
        D, K = self.embedding_dim, self.num_embeddings
        flat = inputs.reshape(-1, D)
        x = flat.detach().contiguous()
        cb = codebook.detach().contiguous()
        sel = None
        if thres is not None:
            if roll is None:
                roll = self.draw_roll(individual, x.device)
            thres_t = torch.as_tensor(thres, dtype=torch.float32, device=x.device)
            sel = (roll.to(x.device) >= thres_t).to(torch.float32).expand(1, K).reshape(K).contiguous()
        idx, quant, dist = _C.vq_assign(x, cb, sel_mask=sel, want_quant=True, want_dist=return_distances)
        encoding_indices = idx.reshape(inputs.shape[:-1])
        quantized, e_latent_loss = _SteAndCommitment.apply(inputs, quant)        # straight-through estimator + mean((sg(q) - x)^2)
        ret = {}
        counts = None
        if is_training:
            counts, dw = _C.vq_ema_stats(x, idx, K)
            local_counts = counts
            if self.stats_all_reduce is not None:
                counts, dw = self.stats_all_reduce(counts, dw)
            if self.fuse_ema_update and cb.dtype == torch.float32 and self.ema_dw.hidden.is_cuda and K <= 1024 and cb.is_contiguous():
                # both moving averages, the Laplace-smoothed cluster sizes and the codebook move in ONE launch (vqn_vq_ema_update)
                ret['update'] = _C.vq_ema_update(counts.contiguous(), dw.contiguous(), cb, self.decay, self.epsilon,
                                                 self.ema_cluster_size, self.ema_dw)
            else:
                cs = self.ema_cluster_size(counts)
                ema_dw = self.ema_dw(dw)
                n = cs.sum()
                cs = (cs + self.epsilon) / (n + K * self.epsilon) * n
                w = ema_dw / cs.reshape(1, -1)
                used = (counts > 0).to(w.dtype)
                ret['update'] = w * used[None, :] + cb * (1.0 - used[None, :])
            counts = local_counts
        else:
            counts = _C.vq_counts(idx, K)
        loss = self.commitment_cost * e_latent_loss

        def perplexity():
            avg_probs = counts / max(idx.numel(), 1)              # = mean(encodings, 0) of this replica's rows
            return torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-10)))
        ret.update({'quantize': quantized, 'loss': loss, 'encoding_indices': encoding_indices, 'distances': dist})
        # `encodings` (an [N, K] one-hot) and `perplexity` are computed when somebody reads them (nobody does in a training step)
        return LazyResult(ret, {'perplexity': perplexity, 'encodings': lambda: torch.nn.functional.one_hot(idx, K).to(flat.dtype)})

    def train_from_raw(self, z_raw, codebook, thres=None, individual=True, roll=None, eps=1e-6):
        """Training form taking the UN-normalised encoder output [N, D] (device, D % 4 == 0, D <= 256): the chain of `forward(
        l2_normalize_rows(z_raw), codebook, is_training=True, ...)` with the normalise / assign / straight-through / commitment part as one
        pass (`_QuantiseTrain`), then the EMA statistics and codebook move as before.  Same result keys (`distances` on first access)."""
        D, K = self.embedding_dim, self.num_embeddings
        cb = codebook.detach().contiguous()
        sel = None
        if thres is not None:
            if roll is None:
                roll = self.draw_roll(individual, z_raw.device)
            thres_t = torch.as_tensor(thres, dtype=torch.float32, device=z_raw.device)
            sel = (roll.to(z_raw.device) >= thres_t).to(torch.float32).expand(1, K).reshape(K).contiguous()
        aux = {}
        quantized, commitment = _QuantiseTrain.apply(z_raw.reshape(-1, D), cb, sel, float(eps), aux, float(self.commitment_cost))
        idx, x = aux['idx'], aux['xnorm']
        counts, dw = _C.vq_ema_stats(x, idx, K)
        local_counts = counts
        ret = {}
        if self.stats_all_reduce is not None:
            counts, dw = self.stats_all_reduce(counts, dw)
        if self.fuse_ema_update and cb.dtype == torch.float32 and self.ema_dw.hidden.is_cuda and K <= 1024 and cb.is_contiguous():
            ret['update'] = _C.vq_ema_update(counts.contiguous(), dw.contiguous(), cb, self.decay, self.epsilon,
                                             self.ema_cluster_size, self.ema_dw)
        else:
            cs = self.ema_cluster_size(counts)
            ema_dw = self.ema_dw(dw)
            n = cs.sum()
            cs = (cs + self.epsilon) / (n + K * self.epsilon) * n
            w = ema_dw / cs.reshape(1, -1)
            used = (counts > 0).to(w.dtype)
            ret['update'] = w * used[None, :] + cb * (1.0 - used[None, :])
        counts = local_counts

        def perplexity():
            avg = counts / max(idx.numel(), 1)
            return torch.exp(-torch.sum(avg * torch.log(avg + 1e-10)))
        ret.update({'quantize': quantized.reshape(z_raw.shape), 'loss': commitment,          # (= commitment_cost * e_latent_loss, from the kernel)
                    'encoding_indices': idx.reshape(z_raw.shape[:-1])})
        return LazyResult(ret, {'perplexity': perplexity, 'encodings': lambda: torch.nn.functional.one_hot(idx, K).to(torch.float32),
                                'distances': lambda: _C.vq_assign(x, cb, sel_mask=sel, want_quant=False, want_dist=True)[2]})

    @torch.no_grad()
    def infer_from_raw(self, z_raw, codebook, thres=None, individual=True, roll=None, eps=1e-6):
        """Inference-only form taking the UN-normalised encoder output: l2-normalise, nearest code, straight-through output,
        commitment term and code usage in ONE kernel (`vqn_vq_quantize_rows`; vq_nfr.py:575-578 + :277-302, :327-330 of this
        file's reference).  Same result keys as `forward(l2_normalize_rows(z_raw), codebook, is_training=False, ...)` and
        bit-identical `quantize` / `encoding_indices`; `encodings`, `distances`, `perplexity` are computed on first access."""
        D, K = self.embedding_dim, self.num_embeddings
        x = z_raw.detach().reshape(-1, D).float().contiguous()
        cb = codebook.detach().contiguous()
        sel = None
        if thres is not None:
            if roll is None:
                roll = self.draw_roll(individual, x.device)
            thres_t = torch.as_tensor(thres, dtype=torch.float32, device=x.device)
            sel = (roll.to(x.device) >= thres_t).to(torch.float32).expand(1, K).reshape(K).contiguous()
        idx, ste, e_latent_loss, counts = _C.vq_quantize_rows(x, cb, sel_mask=sel, eps=eps)

        def perplexity():
            avg = counts / max(idx.numel(), 1)
            return torch.exp(-torch.sum(avg * torch.log(avg + 1e-10)))

        def distances():
            return _C.vq_assign(_C.l2_normalize_rows(x, eps), cb, sel_mask=sel, want_quant=False, want_dist=True)[2]
        return LazyResult({'quantize': ste.reshape(z_raw.shape), 'loss': self.commitment_cost * e_latent_loss,
                           'encoding_indices': idx.reshape(z_raw.shape[:-1])},
                          {'perplexity': perplexity, 'distances': distances,
                           'encodings': lambda: torch.nn.functional.one_hot(idx, K).to(torch.float32)})

    def quantize(self, codebook, encoding_indices):
        return codebook.t()[encoding_indices]


class VectorQuantizer(torch.nn.Module):
    """The gradient-trained VQ layer of vq_layers.py:17-171 (van den Oord et al.; no model of the reference uses it, the EMA
    layer above is the shipped one): nearest code by the same `vqn_vq_assign` kernel; the codebook receives gradients through
    the gathered `quantized` (torch), the encoder through the straight-through estimator.  Per-row losses [N] as in the
    reference: `qloss = mean((q - sg(x))^2, -1)`, `eloss = beta * mean((sg(q) - x)^2, -1)`, `loss = qloss + eloss`."""

    def __init__(self, embedding_dim, num_embeddings, commitment_cost, seed, dtype=torch.float32, name='vector_quantizer'):
        super().__init__()
        self.embedding_dim, self.num_embeddings, self.commitment_cost, self.name = embedding_dim, num_embeddings, commitment_cost, name
        self._gen = torch.Generator(device='cpu')
        self._gen.manual_seed(int(seed))

    def forward(self, inputs, codebook, is_training, thres=None, roll=None):
        D, K = self.embedding_dim, self.num_embeddings
        flat = inputs.reshape(-1, D)
        x = flat.detach().contiguous()
        sel = None
        if thres is not None:
            if roll is None:
                roll = torch.rand((1, K), generator=self._gen)
            thres_t = torch.as_tensor(thres, dtype=torch.float32, device=x.device)
            sel = (roll.to(x.device) >= thres_t).to(torch.float32).expand(1, K).reshape(K).contiguous()
        idx, _, dist = _C.vq_assign(x, codebook.detach().contiguous(), sel_mask=sel, want_quant=False, want_dist=True)
        encodings = torch.nn.functional.one_hot(idx, K).to(flat.dtype)
        encoding_indices = idx.reshape(inputs.shape[:-1])
        quantized = self.quantize(codebook, encoding_indices)
        e_latent = ((quantized.detach() - inputs) ** 2).mean(-1)
        q_latent = ((quantized - inputs.detach()) ** 2).mean(-1)
        if self.commitment_cost > 0:
            e_w = self.commitment_cost * e_latent
            loss = q_latent + e_w
        else:
            loss, e_w = q_latent, torch.zeros_like(q_latent)
        quantized = inputs + (quantized - inputs).detach()
        avg_probs = _C.vq_counts(idx, K) / max(idx.numel(), 1)
        perplexity = torch.exp(-torch.sum(avg_probs * torch.log(avg_probs + 1e-10)))
        return {'quantize': quantized, 'loss': loss, 'qloss': q_latent, 'eloss': e_w, 'perplexity': perplexity, 'encodings': encodings,
                'encoding_indices': encoding_indices, 'distances': dist}

    def quantize(self, codebook, encoding_indices):
        return codebook.t()[encoding_indices]
