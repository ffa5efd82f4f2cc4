"""VQ-stage trainer pieces: mirror of decomp/nerfvq_nfr3/nerfactor/train_nfr.py -- optimiser construction
(:121-139), `outer_sample` (:380-467), `train_iter` / `vali_iter` / `vali_vq` (:562-594) -- on the MI355X model
classes, with rank-sharded data parallelism added (the reference's VQ stage is single-device; its only DP site is
trainvali.py:436-486 for stages 1/3, which `train_iter` covers as well since the step contract is the same).

The view loader is datasets/shape_unit.py; out of scope here: checkpoint managers, TensorBoard.
"""
import torch

from vqnerf_release_amd import parallel


def make_optimizer(config, params, capturable=False):
    """Keras Adam(lr, amsgrad=True) (train_nfr.py:121-139): epsilon 1e-7 (Keras default, torch's is 1e-8), optional
    ExponentialDecay(lr_decay_steps, lr_decay_rate) -> returned as a LambdaLR, clipnorm / clipvalue as a closure.
    `capturable`: step counters and lr live on the device, as `Trainer(graph=True)` needs (and the update is torch's fused
    multi-tensor kernel: one launch instead of a dozen foreach passes)."""
    lr = config.getfloat('DEFAULT', 'lr')
    params = list(params)
    if capturable:
        lr = torch.tensor(lr, dtype=torch.float32, device=params[0].device)
    opt = torch.optim.Adam(params, lr=lr, eps=1e-7, amsgrad=True, capturable=capturable, fused=bool(capturable))
    decay_steps = config.getint('DEFAULT', 'lr_decay_steps', fallback=-1)
    sched = None
    if decay_steps > 0:
        rate = config.getfloat('DEFAULT', 'lr_decay_rate')
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda step: rate ** (step / decay_steps))
    clipnorm = config.getfloat('DEFAULT', 'clipnorm', fallback=-1)
    clipvalue = config.getfloat('DEFAULT', 'clipvalue', fallback=-1)
    assert not (clipnorm > 0 and clipvalue > 0), 'Both `clipnorm` and `clipvalue` are active -- turn one off'
    return opt, sched, (clipnorm, clipvalue)


def compute_average_loss(per_example_loss, global_batch_size):
    """tf.nn.compute_average_loss: sum(per_example) / global_batch_size (NOT the local mean)."""
    return per_example_loss.sum() / global_batch_size


class Trainer:
    """Holds the DP plumbing of one model: flat gradient bucket + VQ statistics reducer.

    `graph=True` (single rank): the step at the reference's batch size (n_rays_per_step = 1024 pairs) is ~150 short
    launches, i.e. launch-bound; the whole of it -- forward, loss, backward, EMA codebook move, Adam -- is captured once
    into a HIP graph and replayed on static buffers.  Conditions, all checked: batches of a fixed shape whose rows are all
    foreground (`outer_sample` only yields such rows; sets `model.assume_foreground`), no `thres` / `roll`, a `capturable`
    optimiser (make_optimizer(..., capturable=True)).  The first `GRAPH_WARMUP` calls run eagerly (they are real steps);
    the tensors returned by later calls are the graph's static outputs, overwritten by the next call."""

    GRAPH_WARMUP = 2

    def __init__(self, model, optimizer, clip=(-1, -1), sched=None, graph=False):
        assert model.trainable_registered, 'Register the trainable layers before using `trainable_variables`'
        self.model, self.optimizer, self.clip, self.sched = model, optimizer, clip, sched
        self.bucket = None
        self.graph = bool(graph)
        self._calls, self._captured, self._static_in, self._static_out = 0, None, None, None
        if self.graph:
            if parallel.is_dist():
                raise RuntimeError('Trainer(graph=True) is single-rank: the captured step holds no collective')
            if not all(g.get('capturable', False) for g in optimizer.param_groups):
                raise ValueError('Trainer(graph=True) needs a capturable optimiser (make_optimizer(..., capturable=True))')
            if sched is not None and not all(torch.is_tensor(g['lr']) for g in optimizer.param_groups):
                raise ValueError('a scheduler under Trainer(graph=True) needs a tensor learning rate')
            model.assume_foreground = True
        # lazily created variables (light, gamma, codebook) must exist before the gradient bucket is laid out
        _ = model.light
        if getattr(model, 'data_type', 'nerf') != 'nerf':
            _ = model.gamma
        if hasattr(model, 'vq_layer'):
            model.get_codebook()
            model.vq_layer.stats_all_reduce = parallel.VQStatsReducer()

    def train_iter(self, batch, global_bs, thres=None, roll=None):
        """One step (train_nfr.py:562-576).  `global_bs` is the reference's normaliser (n_rays_per_step, :571-572) times
        the number of ranks when each rank draws its own rays.  Returns (weighted_loss summed over ranks, to_vis, loss_dict)."""
        if self.graph:
            if thres is not None or roll is not None:
                raise ValueError('code dropout (`thres` / `roll`) is drawn on the host: not available under graph=True')
            self._calls += 1
            if self._calls > self.GRAPH_WARMUP:
                return self._replay(batch, global_bs)
        return self._step(batch, global_bs, thres, roll)

    def _replay(self, batch, global_bs):
        if self._captured is None:
            self._static_in = [t.clone() if torch.is_tensor(t) else t for t in batch]
            self._global_bs = global_bs
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._static_out = self._step(tuple(self._static_in), global_bs, None, None, sched=False, fresh_leaves=True)
            self._captured = g
        if global_bs != self._global_bs:
            raise ValueError('global_bs is baked into the captured step')
        for dst, src in zip(self._static_in, batch):
            if torch.is_tensor(dst):
                if dst.shape != src.shape:
                    raise ValueError(f'captured step takes batches of shape {tuple(dst.shape)}, got {tuple(src.shape)}')
                dst.copy_(src)
        self._captured.replay()
        if self.sched is not None:
            self.sched.step()
        return self._static_out

    def _step(self, batch, global_bs, thres, roll, sched=True, fresh_leaves=False):
        model = self.model
        if self.bucket is None:
            self.bucket = parallel.FlatBucket(model.trainable_variables, n_extra=1)
        if fresh_leaves:
            # (capture only) run the step on fresh leaf aliases of the parameters.  A parameter's AccumulateGrad node is bound to the
            # stream the parameter was first used on and lives as long as ANY tensor derived from it (a caller holding `model.light`,
            # say): under capture the engine would synchronise the capture stream with that old (default) stream, which pulls it
            # into the capture and crashes hipStreamEndCapture.  Fresh leaves share the storage, get their nodes on the capture
            # stream, and their gradients are copied into the bucket the optimiser reads.
            from torch.nn.utils.stateless import _reparametrize_module
            names = {id(p): n for n, p in model.named_parameters()}
            leaves = [p.detach().requires_grad_(True) for p in self.bucket.params]
            with _reparametrize_module(model, {names[id(p)]: q for p, q in zip(self.bucket.params, leaves)}):
                return self._step_body(batch, global_bs, thres, roll, sched, leaves)
        return self._step_body(batch, global_bs, thres, roll, sched, None)

    def _step_body(self, batch, global_bs, thres, roll, sched, leaves):
        model = self.model
        self.optimizer.zero_grad(set_to_none=True)
        self.bucket.attach()
        kw = {'thres': thres}
        if roll is not None:
            kw['roll'] = roll
        pred, gt, loss_kwargs, to_vis = model(batch, mode='train', **kw) if hasattr(model, 'vq_layer') else model(batch, mode='train')
        loss_kwargs.pop('pretrain', None), loss_kwargs.pop('env', None)
        per_example, loss_dict = model.compute_loss(pred, gt, **loss_kwargs)
        weighted = compute_average_loss(per_example, global_bs)
        if leaves is None:
            weighted.backward()
        else:
            grads = torch.autograd.grad(weighted, leaves, allow_unused=True)
            with torch.no_grad():
                for v, g in zip(self.bucket.views, grads):
                    if g is None:
                        v.zero_()
                    else:
                        v.copy_(g)
        with torch.no_grad():
            self.bucket.extra[0] = weighted
        extra = self.bucket.all_reduce()
        clipnorm, clipvalue = self.clip
        if clipnorm > 0:
            # Keras clips every gradient tensor by its own norm
            for p in self.bucket.params:
                n = p.grad.norm()
                p.grad.mul_(torch.clamp(clipnorm / (n + 1e-12), max=1.0))
        if clipvalue > 0:
            self.bucket.flat[:self.bucket.n_grad].clamp_(-clipvalue, clipvalue)
        self.optimizer.step()
        if sched and self.sched is not None:
            self.sched.step()
        # hand back values, not autograd history: a caller holding on to the loss would keep this step's AccumulateGrad nodes
        # (and their stream) alive into the next one, which is what breaks a later capture
        det = lambda d: {k: (v.detach() if torch.is_tensor(v) else v) for k, v in d.items()}
        return extra[0], det(to_vis), det(loss_dict)


def train_iter(model, batch, optimizer, global_bs, thres=None, _trainers={}):
    """Function form with the reference's signature (train_nfr.py:562)."""
    key = (id(model), id(optimizer))
    if key not in _trainers:
        _trainers[key] = Trainer(model, optimizer)
    return _trainers[key].train_iter(batch, global_bs, thres=thres)


@torch.no_grad()
def vali_iter(model, batch, global_bs, thres=None, full_vis=False):
    pred, gt, loss_kwargs, to_vis = model(batch, mode='vali', thres=thres, full_vis=full_vis)
    per_example, loss_dict = model.compute_loss(pred, gt, **loss_kwargs)
    return compute_average_loss(per_example, global_bs), to_vis, loss_dict


@torch.no_grad()
def vali_vq(model, batch, thres=None, full_vis=False):
    pred, gt, loss_kwargs, _ = model.vq_test(batch, mode='vali', thres=thres)
    _, loss_dict = model.compute_loss(pred, gt, **loss_kwargs)
    return loss_dict


@torch.no_grad()
def outer_sample(batch, config, data_type, alpha_thres=0.9, generator=None):
    """Pair sampler (train_nfr.py:380-467): `n_rays_per_step` interior foreground pixels, each with one random
    8-neighbour, interleaved [p1, p1_n, p2, p2_n, ...] -- entirely on the device (the reference syncs `hw[0,:]`
    to the host and gathers with TF ops).  `batch` holds one full view, rays on dim 0 in row-major (h, w) order."""
    bs = config.getint('DEFAULT', 'n_rays_per_step')
    tensors = list(batch)
    id_, hw, alpha = tensors[0], tensors[1], tensors[5]
    H, W = int(hw[0, 0]), int(hw[0, 1])
    dev = alpha.device
    jit = torch.tensor([[-1, -1], [-1, 0], [-1, 1], [0, -1], [0, 1], [1, -1], [1, 0], [1, 1]], device=dev)
    ii, jj = torch.meshgrid(torch.arange(1, H - 1, device=dev), torch.arange(1, W - 1, device=dev), indexing='ij')
    coords = torch.stack([ii, jj], -1).reshape(-1, 2)
    pick = torch.randint(0, 8, (coords.shape[0],), device=dev, generator=generator)
    coords_n = coords + jit[pick]
    a2 = alpha.reshape(H, W)
    if alpha_thres is not None:
        keep = (a2[coords[:, 0], coords[:, 1]] > alpha_thres) & (a2[coords_n[:, 0], coords_n[:, 1]] > alpha_thres)
        coords, coords_n = coords[keep], coords_n[keep]
    sel = torch.randint(0, coords.shape[0], (bs,), device=dev, generator=generator)
    pairs = torch.stack([coords[sel], coords_n[sel]], 1).reshape(-1, 2)           # [p1, p1_n, p2, p2_n, ...]
    flat = pairs[:, 0] * W + pairs[:, 1]
    out = []
    for t in tensors:
        if torch.is_tensor(t):
            out.append(t[flat])
        else:
            # ids: a per-ray list is gathered (host), a per-view id (one element, what datasets.shape_unit yields) passes through
            out.append([t[int(i)] for i in flat.tolist()] if isinstance(t, (list, tuple)) and len(t) > 1 else t)
    return tuple(out)


@torch.no_grad()
def z_cluster(model, init_batch_vis, init_z_path, num_embed, device='cuda', n_samples=None, seed=1):
    """Codebook initialisation (train_nfr.py:471-488): k-means over the encoder's latents of the training views; writes the
    `[num_embed, z_dim]` centres to `init_z_path` (the `cluster_center_path` that vq_nfr.Model.get_codebook loads)."""
    import numpy as np
    from vqnerf_release_amd.decomp.nerfactor.util.torch_kmeans import kmeans
    zs = torch.cat([torch.as_tensor(z) for z in init_batch_vis], 0).to(device)
    if n_samples is not None:
        index = np.array(range(zs.shape[0]))
        np.random.shuffle(index)
        zs = zs[torch.as_tensor(index[:int(n_samples)], device=zs.device)]
    _, centers = kmeans(X=zs, num_clusters=num_embed, distance='euclidean', device=device, seed=seed)
    z_centers = centers.detach().cpu().numpy()
    if init_z_path:
        np.save(init_z_path, z_centers)
    return z_centers
