"""Data parallelism for the two trainers: one process per GPU, `torch.distributed` over RCCL (backend "nccl" on
ROCm) -- gloo on CPU for tests.  Replaces the only DP site of the reference, `tf.distribute.MirroredStrategy`
(decomp/nerfvq_nfr3/nerfactor/trainvali.py:436-446,450-486,518-535), and adds DP for the VQ stage, which the
reference runs single-device (train_nfr.py:562-576).

Design: every message of a step is small (<= 6 MB: 1.4 M geo parameters, 0.78 M decomp parameters,
(D+1)*K codebook statistics, a few scalars), so the step is latency-bound on xGMI.  Hence ONE flat fp32 bucket per
step -- [grads || extras (loss, n_fg, ...)] -- and ONE all-reduce(sum); the VQ layer's EMA statistics
[counts (K) || dw (D*K)] need their reduction in the middle of the forward pass (the codebook update and the
`used` mask depend on the global counts) and go in a second, tiny all-reduce.  After the reduce every rank applies
the identical EMA update and optimiser step, so weights and codebooks stay bit-identical across ranks without any
broadcast.  Rays / surface points are sharded by rank with no data-path collective.
"""
import torch
import torch.distributed as dist


def _clock(name):
    """Device-time bracket of a collective (HIP events on the launch stream, `_C.KernelClock`; a no-op unless switched on)."""
    from vqnerf_release_amd import _C
    return _C._clock(name)


_capture = None          # the SegmentedCapture that is recording, if any


def all_reduce_sum(t, group=None, what='all_reduce:scalars'):
    """dist.all_reduce(SUM) with the optional device-time bracket (bench.py reports collective time per step).  While a
    SegmentedCapture records, the collective is not issued: the graph segment ends here, the tensor is noted as the exchange
    between this segment and the next, and a new segment begins."""
    if _capture is not None:
        _capture.cut(t, group, what)
        return t
    with _clock(what):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


class SegmentedCapture:
    """A training step as HIP graphs WITH its collectives: every `all_reduce_sum` met while recording ends the current graph and
    starts the next, so a step becomes  graph 0 -> all-reduce(buffer 0) -> graph 1 -> all-reduce(buffer 1) -> graph 2 ...  and
    `replay()` launches the graphs with the eager collectives between them.  The exchanged tensors are static buffers (the flat
    gradient bucket, the VQ statistics buffer), all segments share one memory pool, and the recording runs on one side stream.
    With a single rank nothing cuts and the step is one graph.

        cap = SegmentedCapture()
        with cap:
            out = step(static_inputs)          # recorded, not executed
        cap.replay()                           # every later step
    """

    def __init__(self):
        self.graphs, self.exchanges = [], []
        self._pool = None
        self._stream = None
        self._ctx = None

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()       # one memory pool for every segment: tensors live across the cuts
        g.capture_begin(pool=self._pool)
        self.graphs.append(g)

    def __enter__(self):
        import gc
        global _capture
        assert _capture is None, 'captures do not nest'
        torch.cuda.synchronize()
        gc.collect()
        self._stream = torch.cuda.Stream()
        self._stream.wait_stream(torch.cuda.current_stream())
        self._ctx = torch.cuda.stream(self._stream)
        self._ctx.__enter__()
        self._begin()
        _capture = self
        return self

    def cut(self, tensor, group, what):
        self.graphs[-1].capture_end()
        self.exchanges.append((tensor, group, what))
        self._begin()

    def __exit__(self, exc_type, exc, tb):
        global _capture
        _capture = None
        try:
            if exc_type is None:
                self.graphs[-1].capture_end()
            else:
                # the body raised: end the capture only to leave the stream usable, and never let a failure of THAT mask the
                # original error
                try:
                    self.graphs[-1].capture_end()
                except Exception:
                    pass
                self.graphs = []
        finally:
            self._ctx.__exit__(exc_type, exc, tb)
            torch.cuda.current_stream().wait_stream(self._stream)
        return False

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.exchanges):
                t, group, what = self.exchanges[i]
                with _clock(what):
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def world_size():
    return dist.get_world_size() if is_dist() else 1


def backend():
    """'nccl' (= RCCL on ROCm) / 'gloo' of the default process group, None without one."""
    return dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def multi_copy(dsts, srcs):
    """dsts[i].copy_(srcs[i]): ONE HIP launch for all pairs that are contiguous f32 device tensors (vqn_multi_copy -- a launch per
    parameter is ~60 of a captured 2048-point training step's ~420 launches), torch's foreach copy for the others (an expanded or
    transposed gradient)."""
    fast = [d.is_cuda and s.is_cuda and d.is_contiguous() and s.is_contiguous() and d.dtype == s.dtype == torch.float32 and
            d.numel() == s.numel() for d, s in zip(dsts, srcs)]
    if any(fast):
        from vqnerf_release_amd import _C
        _C.multi_copy([d for d, f in zip(dsts, fast) if f], [s for s, f in zip(srcs, fast) if f])
    rest = [(d, s) for d, s, f in zip(dsts, srcs, fast) if not f]
    if rest:
        torch._foreach_copy_([d for d, _ in rest], [s.reshape(d.shape) for d, s in rest])


class FlatBucket:
    """Persistent flat fp32 buffer viewed as the gradients of `params` followed by `n_extra` scalars."""

    def __init__(self, params, n_extra=0):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        self.n_extra = n_extra
        dev = self.params[0].device
        self.sizes = [p.numel() for p in self.params]
        self.n_grad = sum(self.sizes)
        self.flat = torch.zeros(self.n_grad + n_extra, dtype=torch.float32, device=dev)
        self.views, o = [], 0
        for p, n in zip(self.params, self.sizes):
            self.views.append(self.flat[o:o + n].view_as(p))
            o += n
        self.extra = self.flat[self.n_grad:]

    def attach(self, zero=True):
        """Make every p.grad a view into the bucket: backward then writes straight into it (no pack copy).  `zero=False`: the caller
        fills every view itself (gradients as values + one multi-tensor copy, zeros for parameters without one): the 11 MB fill of the
        bucket is then one launch too many."""
        if zero:
            self.flat.zero_()
        for p, v in zip(self.params, self.views):
            p.grad = v
        self._reset_overlap()
        return self

    def _reset_overlap(self):
        """Forget slices in flight (waiting for them first): a step that raised between backward and all_reduce(), or a second
        backward before the reduce, must not leave stale work handles or a half-counted slice behind."""
        if getattr(self, '_slices', None) is not None:
            for w in self._pending.values():
                w.wait()
            self._pending, self._left = {}, [sl['n_params'] for sl in self._slices]

    def disable_overlap(self):
        """Remove the post-accumulate hooks (a second bucket / Trainer on the same model would otherwise fire this one's collectives)."""
        for h in getattr(self, '_hooks', []):
            h.remove()
        self._hooks = []
        if getattr(self, '_slices', None) is not None:
            self._reset_overlap()
        self._slices = None

    def __del__(self):
        try:
            for h in getattr(self, '_hooks', []):
                h.remove()
        except Exception:
            pass

    def gather_grads(self):
        """For parameters whose .grad is not (or no longer) a view of the bucket."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v

    # ---- overlap with the backward tail (trainvali.py:469-477 lets MirroredStrategy overlap its all-reduces the same way) ----
    def enable_overlap(self, n_buckets=2, group=None):
        """Cut the gradient part of the buffer into `n_buckets` contiguous slices (by bytes, along the parameter order) and send
        each slice off (asynchronous all-reduce: RCCL runs it on its own stream, ordered after the kernels already queued on the
        launch stream) as soon as the LAST gradient of the slice has been accumulated, while the backward pass goes on producing the
        others.  The backward pass reaches the parameters roughly in reverse order, so the slices complete back to front; the
        extras (loss terms, written after backward) travel with whatever is reduced at `all_reduce()`, which also reduces every
        slice whose hooks did not all fire (an unused parameter).  Sums are slice-local, element-wise: the result is bit-identical
        to the single all-reduce.  Not used while a SegmentedCapture records (a collective cuts the graph there).
        CONTRACT: every rank must run the same autograd graph -- the slices go out in hook-firing order, and a parameter that is unused
        on ONE rank only would make the ranks' collective sequences differ (a hang).  Callers that cannot promise that keep the single
        all-reduce (do not call this).  One backward per all_reduce(): a second backward that reaches a slice already in flight (gradient
        accumulation) raises -- the slice left with the first pass's sums only."""
        if getattr(self, '_slices', None) is not None:
            return self
        target = max(1, -(-self.n_grad // max(1, n_buckets)))
        self._slices, lo, acc, first = [], 0, 0, 0
        self._slice_of = {}
        for i, n in enumerate(self.sizes):
            acc += n
            self._slice_of[id(self.params[i])] = len(self._slices)
            if acc >= target or i == len(self.sizes) - 1:
                self._slices.append({'lo': lo, 'hi': lo + acc, 'n_params': i + 1 - first})
                lo, acc, first = lo + acc, 0, i + 1
        self._group = group
        self._pending, self._left = {}, [sl['n_params'] for sl in self._slices]

        def make_hook(i, k):
            def hook(p):
                if not is_dist() or _capture is not None:
                    return
                v = self.views[i]
                if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                    p.grad = v
                self._left[k] -= 1
                if self._left[k] < 0:                  # a second backward before all_reduce(): the slice is already in flight with
                    w = self._pending.pop(k, None)     # the FIRST pass's sums -- take it back (wait) and let all_reduce() send the
                    if w is not None:                  # accumulated gradients as part of its own run
                        w.wait()
                        raise RuntimeError('FlatBucket.enable_overlap: a second backward reached a slice that had already been sent; '
                                           'accumulate gradients with the single all-reduce (do not enable the overlap)')
                    return
                if self._left[k] == 0:
                    sl = self._slices[k]
                    with _clock('all_reduce:grad_bucket[%d]' % k):
                        self._pending[k] = dist.all_reduce(self.flat[sl['lo']:sl['hi']], op=dist.ReduceOp.SUM, group=group, async_op=True)
            return hook
        self._hooks = [p.register_post_accumulate_grad_hook(make_hook(i, self._slice_of[id(p)])) for i, p in enumerate(self.params)]
        return self

    def all_reduce(self, average_grads=False, group=None):
        """One collective for the whole step (or, after `enable_overlap`, the wait for the slices already in flight + one collective
        for the rest).  Gradients are summed (the trainers normalise their loss by the GLOBAL batch, as train_nfr.py:571-572 does
        with `global_batch_size`), or averaged on request."""
        slices = getattr(self, '_slices', None)
        if slices is None or not is_dist() or _capture is not None:
            self.gather_grads()
            if is_dist():
                all_reduce_sum(self.flat, group, 'all_reduce:grad_bucket')
                if average_grads:
                    self.flat[:self.n_grad].div_(dist.get_world_size(group))
            return self.extra
        # parameters that saw no gradient this step: their views must hold zeros before the slice goes out
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            k = self._slice_of[id(p)]
            if k in self._pending:
                continue
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
                p.grad = v
        # slices not yet sent (+ the extras) go out as one contiguous run where possible: the unsent slices are a prefix in the
        # usual back-to-front completion order
        todo = [k for k in range(len(slices)) if k not in self._pending]
        runs, k = [], 0
        while k < len(todo):
            j = k
            while j + 1 < len(todo) and todo[j + 1] == todo[j] + 1:
                j += 1
            runs.append((slices[todo[k]]['lo'], slices[todo[j]]['hi']))
            k = j + 1
        tail = (self.n_grad, self.n_grad + self.n_extra)
        if runs and runs[-1][1] == self.n_grad:
            runs[-1] = (runs[-1][0], tail[1])
        elif self.n_extra:
            runs.append(tail)
        for lo, hi in runs:
            all_reduce_sum(self.flat[lo:hi], group, 'all_reduce:grad_bucket')
        for w in self._pending.values():
            w.wait()                                   # the launch stream waits for the collectives that ran beside the backward tail
        self._pending, self._left = {}, [sl['n_params'] for sl in slices]
        if average_grads:
            self.flat[:self.n_grad].div_(dist.get_world_size(group))
        return self.extra


class VQStatsReducer:
    """`VectorQuantizerEMA.stats_all_reduce` hook: sums [counts || dw] over ranks in one small all-reduce."""

    def __init__(self, group=None):
        self.group = group
        self._buf = None

    def __call__(self, counts, dw):
        if not is_dist():
            return counts, dw
        K, n = counts.numel(), counts.numel() + dw.numel()
        if self._buf is None or self._buf.numel() != n or self._buf.device != counts.device:
            self._buf = torch.empty(n, dtype=torch.float32, device=counts.device)
        self._buf[:K].copy_(counts.reshape(-1))
        self._buf[K:].copy_(dw.reshape(-1))
        all_reduce_sum(self._buf, self.group, 'all_reduce:vq_stats')
        return self._buf[:K].clone().view_as(counts), self._buf[K:].clone().view_as(dw)


def shard_range(n, r=None, w=None):
    """Contiguous [lo, hi) share of n independent units (rays, views, surface points) for rank r of w."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    base, rem = divmod(n, w)
    lo = r * base + min(r, rem)
    return lo, lo + base + (1 if r < rem else 0)


def broadcast_module(module, src=0):
    """Start-up only (e.g. after rank 0 ran the k-means codebook init): make every rank's state identical."""
    if not is_dist():
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)


def assert_replicas_identical(tensors, what='parameters'):
    """Debug check used by tests: every rank holds bit-identical tensors."""
    if not is_dist():
        return True
    for t in tensors:
        ref = t.detach().clone()
        dist.broadcast(ref, src=0)
        if not torch.equal(ref, t.detach()):
            raise AssertionError(f'{what} diverged across ranks')
    return True
