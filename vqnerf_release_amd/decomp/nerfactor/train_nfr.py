"""VQ-stage trainer pieces: mirror of decomp/nerfvq_nfr3/nerfactor/train_nfr.py -- optimiser construction
(:121-139), `outer_sample` (:380-467), `train_iter` / `vali_iter` / `vali_vq` (:562-594) -- on the MI355X model
classes, with rank-sharded data parallelism added (the reference's VQ stage is single-device; its only DP site is
trainvali.py:436-486 for stages 1/3, which `train_iter` covers as well since the step contract is the same).

The view loader is datasets/shape_unit.py.  `fit` (bottom of the file) is the epoch loop of the reference's main()
(:93-378): threshold schedule, VQ test set, k-means init, checkpoints, validation output.  Out of scope: absl flags,
tf.train.CheckpointManager, TensorBoard.
"""
import torch

from vqnerf_release_amd import parallel


def make_optimizer(config, params, capturable=False):
    """Keras Adam(lr, amsgrad=True) (train_nfr.py:121-139): epsilon 1e-7 (Keras default) added to the UN-debiased sqrt(vhat) as
    TensorFlow's kernel does (optim.HipAdam(eps_mode='keras'); oracle/optim.py: keras_adam_step), optional
    ExponentialDecay(lr_decay_steps, lr_decay_rate) -> returned as a LambdaLR, clipnorm / clipvalue as a closure.
    `capturable`: the lr lives on the device as well as the step counters, as `Trainer(graph=True)` needs.  On a GPU the update is one
    launch (csrc/adam.hip; torch's fused multi-tensor kernel takes 2 x 85 us for these ~1 M parameters, 16 workgroups each); CPU
    parameters (the gloo tests) take the framework statement of the same update."""
    from vqnerf_release_amd.optim import HipAdam
    # (the f32 value of the configured rate in BOTH execution modes: the captured step reads it from a device float, the eager one hands the
    #  kernel a host double -- 5e-3 is not an f32 number, and the two runs would differ in the last place from the first update on)
    lr = float(torch.tensor(config.getfloat('DEFAULT', 'lr'), dtype=torch.float32))
    params = list(params)
    if capturable:
        lr = torch.tensor(lr, dtype=torch.float32, device=params[0].device)
    opt = HipAdam(params, lr=lr, eps=1e-7, amsgrad=True, eps_mode='keras', capturable=True if capturable else None)
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

    `graph=True`: the step at the reference's batch size (n_rays_per_step = 1024 pairs) is ~150 short
    launches, i.e. launch-bound; the whole of it -- forward, loss, backward, EMA codebook move, Adam -- is captured once
    into a HIP graph and replayed on static buffers.  Conditions, all checked: batches of a fixed shape whose rows are all
    foreground (`outer_sample` only yields such rows; sets `model.assume_foreground`), no explicit `roll` (`thres` as a device
    tensor is a graph input; its draw then comes from the device generator), a `capturable` optimiser (make_optimizer(..., capturable=True)).  The first `GRAPH_WARMUP` calls run eagerly (they are real steps);
    the tensors returned by later calls are the graph's static outputs, overwritten by the next call."""

    GRAPH_WARMUP = 2

    def __init__(self, model, optimizer, clip=(-1, -1), sched=None, graph=False):
        assert model.trainable_registered, 'Register the trainable layers before using `trainable_variables`'
        self.model, self.optimizer, self.clip, self.sched = model, optimizer, clip, sched
        self.bucket = None
        self.graph = bool(graph)
        self._calls, self._captured, self._static_in, self._static_out, self._captured_key = 0, None, None, None, ()
        if self.graph:
            # (data parallel: the captured step is cut at its two collectives -- VQ statistics, gradient bucket -- into three
            # graphs with the eager all-reduces between them: parallel.SegmentedCapture)
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

    def train_iter(self, batch, global_bs, thres=None, roll=None, **call_kwargs):
        """One step (train_nfr.py:562-576; trainvali.py:450-486 for the stage-1 / stage-3 models, whose extra call
        arguments -- `pretrain`, `bias_weight` -- go through `call_kwargs`).  `global_bs` is the reference's normaliser (n_rays_per_step, :571-572) times
        the number of ranks when each rank draws its own rays.  Returns (weighted_loss summed over ranks, to_vis, loss_dict)."""
        self._call_kwargs = call_kwargs
        if self.graph:
            if roll is not None:
                raise ValueError('an explicit `roll` is host-side: not available under graph=True')
            # per-call arguments (`pretrain`, `bias_weight` of the stage-1 / stage-3 models) are host-side constants of the recorded step:
            # a call with other values than the captured step's drops that capture and records again (fit_stage: once, when the
            # pretraining epochs end)
            key = tuple(sorted((k, v if isinstance(v, (bool, int, float, str, type(None))) else id(v)) for k, v in call_kwargs.items()))
            if self._captured is not None and key != self._captured_key:
                self._captured, self._static_in, self._static_out = None, None, None
            self._captured_key = key
            if thres is not None and not (torch.is_tensor(thres) and thres.is_cuda):
                raise ValueError('under graph=True the code-dropout thresholds must be a device tensor (they are a graph input)')
            self._calls += 1
            if self._calls > self.GRAPH_WARMUP:
                return self._replay(batch, global_bs, thres)
        return self._step(batch, global_bs, thres, roll)

    def _replay(self, batch, global_bs, thres=None):
        if self._captured is None:
            self._static_in = [t.clone() if torch.is_tensor(t) else t for t in batch]
            self._static_thres = None if thres is None else thres.detach().clone()
            self._global_bs = global_bs
            cap = parallel.SegmentedCapture()
            with cap:
                # (code dropout: the thresholds are a static input, the draw comes from the device generator inside the graph)
                self._static_out = self._step(tuple(self._static_in), global_bs, self._static_thres, None, sched=False, fresh_leaves=True)
            self._captured = cap
        if global_bs != self._global_bs:
            raise ValueError('global_bs is baked into the captured step')
        if (thres is None) != (self._static_thres is None):
            raise ValueError('the captured step was recorded %s code dropout' % ('without' if self._static_thres is None else 'with'))
        if thres is not None:
            self._static_thres.copy_(thres)
        pairs = []
        for dst, src in zip(self._static_in, batch):
            if torch.is_tensor(dst):
                if dst.shape != src.shape:
                    raise ValueError(f'captured step takes batches of shape {tuple(dst.shape)}, got {tuple(src.shape)}')
                if dst.data_ptr() != src.data_ptr():
                    pairs.append((dst, src))
        if pairs:                                      # the batch into the graph's static inputs: one launch, not one per tensor
            parallel.multi_copy([d for d, _ in pairs], [s for _, s in pairs])
        self._captured.replay()
        self.model.weights_changed()       # a replay moves weights and codebook without bumping any tensor `_version`
        if self.sched is not None:
            self.sched.step()
        return self._static_out

    def _step(self, batch, global_bs, thres, roll, sched=True, fresh_leaves=False):
        model = self.model
        if self.bucket is None:
            self.bucket = parallel.FlatBucket(model.trainable_variables, n_extra=1)
            if parallel.is_dist() and not self.graph:
                # eager data parallelism: the heads' gradients are final while the encoder's backward programs still run -- their
                # slice of the bucket is reduced beside them (the captured step is cut at its collectives instead)
                self.bucket.enable_overlap(n_buckets=2)
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
        by_value = not (leaves is None and (parallel.is_dist() or not self.bucket.flat.is_cuda))      # (the branch below that copies the gradients in)
        self.bucket.attach(zero=not by_value)
        kw = {'thres': thres}
        if roll is not None:
            kw['roll'] = roll
        pred, gt, loss_kwargs, to_vis = model(batch, mode='train', **kw) if hasattr(model, 'vq_layer') \
            else model(batch, mode='train', **getattr(self, '_call_kwargs', {}))
        loss_kwargs.pop('pretrain', None), loss_kwargs.pop('env', None)
        per_example, loss_dict = model.compute_loss(pred, gt, **loss_kwargs)
        weighted = compute_average_loss(per_example, global_bs)
        if leaves is None and (parallel.is_dist() or not self.bucket.flat.is_cuda):
            weighted.backward()                      # (data parallel: the bucket's post-accumulate hooks start its slices' all-reduces)
            by_value = False
        else:
            # the gradients as VALUES, moved into the bucket by one multi-tensor copy: `backward()` accumulates into every
            # parameter's `.grad` view with a launch of its own (52 of them per step)
            if getattr(self, '_one', None) is None or self._one.device != weighted.device:
                self._one = torch.ones((), dtype=weighted.dtype, device=weighted.device)        # the root adjoint, made once (not a fill per step)
            # (round 5, measured and NOT kept: the stacks' contractions forked to a side stream inside the captured step -- 0.538 -> 0.571 ms per
            #  replay: a HIP graph's cross-stream edges cost more than the ~85 us of contractions they take off the critical path;
            #  profiles/r05_refl_step.txt)
            grads = torch.autograd.grad(weighted, leaves if leaves is not None else self.bucket.params, grad_outputs=self._one,
                                        allow_unused=True)
            with torch.no_grad():
                # one multi-tensor copy for all gradients (a launch per parameter is ~40 of the captured step's launches)
                have = [(v, g) for v, g in zip(self.bucket.views, grads) if g is not None]
                for v, g in zip(self.bucket.views, grads):
                    if g is None:
                        v.zero_()
                # (the loss rides along: the bucket's extra slot is one more destination of the same launch)
                parallel.multi_copy([v for v, _ in have] + [self.bucket.extra[0:1]], [g for _, g in have] + [weighted.detach().reshape(1)])
                by_value = True
        if not by_value:
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
def outer_sample(batch, config, data_type, alpha_thres=0.9, generator=None, neighbour='random'):
    """Pair sampler (train_nfr.py:380-467; `neighbour='max_diff'`: the variant of trainvali.py:327-336 used by stages 1 / 3,
    which pairs every pixel with the 8-neighbour whose colour differs most, first one on ties): `n_rays_per_step` interior foreground pixels, each with one random
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
    if neighbour == 'max_diff':
        rgb2 = tensors[4].reshape(H, W, -1)
        nb = coords[None, :, :] + jit[:, None, :]                                          # [8, n, 2]
        diff = (rgb2[nb[..., 0], nb[..., 1]] - rgb2[coords[:, 0], coords[:, 1]][None]).abs().amax(-1)
        pick = torch.argmax((diff == diff.amax(0, keepdim=True)).to(torch.uint8), dim=0)  # first maximum, as tf.argmax
    else:
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


# ------------------------------------------------------------------------------------------------------------------
# Epoch driver of the VQ stage: the body of the reference's main() (train_nfr.py:93-378) as a function.  Kept: the code-
# dropout threshold schedule, the VQ test set, the k-means codebook init at step 0, one optimisation step per training
# view and epoch, `ckpt_period` / `vali_period`, the choice of the main codebook size from the drop-loss curve, the
# validation output tree `vis_vali/epoch{e:09d}/{n | main_n}/batch{b:09d}` and `metas.json`.  Not kept (control plane):
# absl flags, tf.train.CheckpointManager (-> torch.save files `ckpt-{step}.pt`), TensorBoard summaries, the matplotlib plot.

def thres_schedule(config):
    """(train_thres [K], val_thres_list (num_drop + 1 arrays, the last one dropping the most codes), x_list)
    (train_nfr.py:183-197).  `thres_str` is ';'-separated (a ',' would split --config_override)."""
    import numpy as np
    num_embed, num_drop = config.getint('DEFAULT', 'num_embed'), config.getint('DEFAULT', 'num_drop')
    thres_str = config.get('DEFAULT', 'thres_str')
    train_thres = [0.0] * (num_embed - num_drop)
    if thres_str != '-':
        train_thres = train_thres + [float(x) for x in thres_str.split(';')]
    val_thres_list = [np.array([0.0] * (num_embed - i) + [1.0] * i) for i in range(num_drop + 1)]
    val_thres_list.reverse()
    return np.array(train_thres), val_thres_list, list(range(num_embed - num_drop, num_embed + 1))


def select_main_vq(drop_losses, main_thres):
    """Index into val_thres_list of the codebook size to present as the main result (train_nfr.py:312-327): the first
    interior i whose loss is lower than its predecessor's and within `main_thres` of every later one; else the last."""
    n = len(drop_losses)
    for i in range(1, n - 1):
        if drop_losses[i - 1] > drop_losses[i] and all(drop_losses[i] - drop_losses[j] <= main_thres for j in range(i + 1, n)):
            return i
    return n - 1


@torch.no_grad()
def prepare_vq_data(config, per_sample_n, views, data_type, generator=None):
    """`per_sample_n` random foreground rows of one pair sample per training view, concatenated (train_nfr.py:489-541)."""
    cols = None
    for view in views:
        batch = outer_sample(view, config, data_type, generator=generator)
        rows = torch.nonzero(batch[5][:, 0] > 0)[:, 0]
        sel = rows[torch.randint(0, rows.numel(), (per_sample_n,), device=rows.device, generator=generator)]
        picked = [t[sel] if torch.is_tensor(t) else t for t in batch]
        if cols is None:
            cols = [[p] for p in picked]
        else:
            for c, p in zip(cols, picked):
                c.append(p)
    return tuple(torch.cat(c, 0) if torch.is_tensor(c[0]) else c[0] for c in cols)


def save_metas(outdir):
    """vis_vali/metas.json: per-epoch means of the per-view metadata metrics (train_nfr.py:387-409)."""
    import json, os
    import numpy as np
    root = os.path.join(outdir, 'vis_vali')
    keys = ('psnr', 'ssim', 'lpips', 'psnr_luma', 'ssim_luma', 'mse')
    metrics = {k: [] for k in keys}
    for e_dir in sorted(os.listdir(root)) if os.path.isdir(root) else []:
        if not e_dir.startswith('epoch'):
            continue
        ep = {k: [] for k in keys}
        for sub, _, files in os.walk(os.path.join(root, e_dir)):
            if os.path.basename(sub).startswith('batch') and 'metadata.json' in files:
                with open(os.path.join(sub, 'metadata.json')) as f:
                    for k, v in json.load(f).items():
                        if k in ep:
                            ep[k].append(v)
        for k in keys:
            metrics[k].append(float(np.mean(ep[k])) if ep[k] else None)
    with open(os.path.join(root, 'metas.json'), 'w') as f:
        json.dump(metrics, f)
    return metrics


def _latest_checkpoint(ckptdir):
    import os, re
    best = None
    for n in os.listdir(ckptdir) if os.path.isdir(ckptdir) else []:
        m = re.fullmatch(r'ckpt-(\d+)\.pt', n)
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), os.path.join(ckptdir, n))
    return best


def graph_default(device, model=None):
    """`graph=None` of fit() / fit_stage(): the captured step (bit-identical to the eager one, ~6x faster at the reference batch) whenever
    it is eligible -- a GPU, the HIP training backend, and a single rank; data-parallel runs keep the explicit opt-in documented at
    fit() (gloo rehearsals: on).  VQN_FIT_GRAPH=0 | 1 overrides."""
    import os
    force = os.environ.get('VQN_FIT_GRAPH')
    if force in ('0', '1'):
        return force == '1'
    if torch.device(device).type != 'cuda' or not torch.cuda.is_available():
        return False
    if model is not None and getattr(model, 'train_backend', 'hip') != 'hip':
        return False
    if model is not None and getattr(model, 'check_numerics', False):
        return False                                   # (debug mode: the numerics guards read a flag back per call -- not recordable)
    return True


def fit(config, outdir, dataset_train, dataset_vali=None, model=None, device='cuda', epochs=None, graph=None, seed=None,
        log=print):
    """Train the VQ stage for `epochs` (config `epochs`) passes over the training views; returns (model, history).
    `graph`: None (default) = the captured step whenever it is eligible (`graph_default`), True / False force it.

    history: {'loss': [mean step loss per epoch], 'vali': [{'step', 'drop_losses', 'main_vq', 'vis_dirs'} ...]}."""
    import json, os
    import numpy as np
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    g = lambda k, cast, fb: cast(config.get('DEFAULT', k, fallback=str(fb)))
    data_type = config.get('DEFAULT', 'data_type')
    seed = g('random_seed', int, 0) if seed is None else seed
    torch.manual_seed(seed)
    # data parallel: every rank sees all views and draws its OWN pairs (generator seeded per rank); what must agree across
    # ranks -- the VQ test set, the k-means codebook -- comes from a generator with the common seed; files are rank 0's
    rank0 = parallel.rank() == 0
    gen_common = torch.Generator(device=device)
    gen_common.manual_seed(seed)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 7919 * parallel.rank())
    os.makedirs(outdir, exist_ok=True)
    global_bs = dataset_train.bs * parallel.world_size()
    views_train = list(dataset_train.build_pipeline(no_shuffle=True))
    vq_test_batch = prepare_vq_data(config, max(1, g('total_sample_vq', int, 4096) // max(1, dataset_train.get_n_views())),
                                    views_train, data_type, generator=gen_common)
    vali_views = []
    if dataset_vali is not None and dataset_vali.get_n_views() > 0:
        vali_views = list(dataset_vali.build_pipeline())[:g('vali_batches', int, 4)]
    if model is None:
        model = get_model_class(config.get('DEFAULT', 'model'))(config)
        model.build_nets(device=device, seed=seed).to(device)
    train_thres, val_thres_list, _ = thres_schedule(config)
    num_embed, num_drop = config.getint('DEFAULT', 'num_embed'), config.getint('DEFAULT', 'num_drop')
    epochs = config.getint('DEFAULT', 'epochs') if epochs is None else epochs
    ckpt_period, vali_period = g('ckpt_period', int, 100), g('vali_period', int, 100)
    keep = g('keep_recent_epochs', int, -1)
    vis_view = g('vis_view', int, 0)
    ckptdir = os.path.join(outdir, 'checkpoints')

    # codebook: k-means over the encoder's latents of the training views at step 0 (train_nfr.py:206-228), else the checkpoint
    latest = _latest_checkpoint(ckptdir)
    step = 0
    if latest is None:
        zs = [model.init_z(outer_sample(v, config, data_type, generator=gen_common))['z_pred'] for v in views_train]
        init_z_path = config.get('DEFAULT', 'cluster_center_path', fallback='') or os.path.join(outdir, 'cluster_init.npy')
        model.set_codebook(z_cluster(model, zs, init_z_path, num_embed, device=device, seed=seed))
    elif model._codebook is None:
        model.set_codebook(np.zeros((num_embed, model.z_dim), np.float32))      # placeholder: the checkpoint below holds the values
    _ = model.light                                              # lazy variables exist before the optimiser is built
    model.register_trainable()
    use_graph = graph_default(device, model) if graph is None else bool(graph)      # (data parallel: the captured step is cut at its collectives)
    if use_graph and parallel.world_size() > 1 and parallel.backend() != 'gloo' and not config.getboolean('DEFAULT', 'dp_graph', fallback=False) \
            and os.environ.get('VQN_DP_GRAPH', '0') in ('', '0'):
        # the graph-segment replay of the DP step has been validated bit-identical to the eager one on two gloo ranks sharing a
        # card, never on a multi-GPU RCCL node: explicit opt-in there (`dp_graph = True` in the config or VQN_DP_GRAPH=1)
        log('graph=True under multi-rank RCCL needs dp_graph=True / VQN_DP_GRAPH=1: running the eager data-parallel step')
        use_graph = False
    opt, sched, clip = make_optimizer(config, model.trainable_variables, capturable=use_graph)
    if latest is not None:
        state = torch.load(latest[1], map_location=device, weights_only=False)
        model.load_state_dict(state['net'])
        opt.load_state_dict(state['optimizer'])
        step = int(state['step'])
        log(f'Resumed from step {step}: {latest[1]}')
    trainer = Trainer(model, opt, clip=clip, sched=sched, graph=use_graph)
    thres_arg = None if np.all(train_thres == 0.0) else torch.tensor(train_thres, dtype=torch.float32, device=device)

    history = {'loss': [], 'vali': []}
    for _ in range(step, epochs):
        losses, loss_dicts = [], []
        for view in views_train:
            batch = outer_sample(view, config, data_type, generator=gen)
            loss, _, loss_dict = trainer.train_iter(batch, global_bs, thres=thres_arg)
            losses.append(loss.detach().clone())
            loss_dicts.append({k: v.detach().float().mean() for k, v in loss_dict.items()})
        step += 1
        history['loss'].append(float(torch.stack(losses).mean()))           # one host sync per epoch
        if step % ckpt_period == 0 and rank0:
            os.makedirs(ckptdir, exist_ok=True)
            torch.save({'step': step, 'net': model.state_dict(), 'optimizer': opt.state_dict()}, os.path.join(ckptdir, f'ckpt-{step}.pt'))
            if keep > 0:
                olds = sorted(int(n[5:-3]) for n in os.listdir(ckptdir) if n.startswith('ckpt-') and n.endswith('.pt'))
                for s in olds[:-keep]:
                    os.remove(os.path.join(ckptdir, f'ckpt-{s}.pt'))
            log(f'Checkpointed step {step}: loss_train {history["loss"][-1]:.6f}')
        if vali_views and vali_period > 0 and step % vali_period == 0 and rank0:
            edir = os.path.join(outdir, 'vis_vali', 'epoch{e:09d}'.format(e=step))
            os.makedirs(edir, exist_ok=True)
            sums = {}
            for d in loss_dicts:
                for k, v in d.items():
                    sums[k] = sums.get(k, 0.0) + float(v)
            with open(os.path.join(edir, 'loss.json'), 'w') as f:
                json.dump(sums, f)
            scores = {'vqrgb': [], 'chromaticity': []}
            for vt in val_thres_list:                            # drop-loss curve on the fixed VQ test set (:285-300)
                ld = vali_vq(model, vq_test_batch, torch.tensor(vt, dtype=torch.float32, device=device))
                scores['vqrgb'].append(float(ld['vqrgb'].mean()))
                scores['chromaticity'].append(float(ld['chromaticity'].mean()) if 'chromaticity' in ld else scores['vqrgb'][-1])
            with open(os.path.join(edir, 'vq_test_loss.json'), 'w') as f:
                json.dump(scores, f)
            main_vq = select_main_vq(scores['chromaticity'], g('best_thres', float, 0.0))
            vis_dirs, writer = [], None
            for i, vt in enumerate(val_thres_list):
                vt_t = torch.tensor(vt, dtype=torch.float32, device=device)
                for b, view in enumerate(vali_views):
                    _, to_vis, _ = vali_iter(model, view, dataset_vali.bs, vt_t, full_vis=(b == vis_view))
                    n_codes = str(num_embed - num_drop + i)
                    vdir = os.path.join(edir, ('main_' + n_codes) if i == main_vq else n_codes, 'batch{b:09d}'.format(b=b))
                    full = os.path.join(outdir, 'vis_vali', 'vis_params', 'epoch{e:09d}'.format(e=step)) \
                        if (b == vis_view and i == len(val_thres_list) - 1) else None
                    writer = model.vis_batch(to_vis, vdir, mode='vali', simp=True, full_vis_path=full)
                    vis_dirs.append(vdir)
            if writer is not None:
                writer.flush()
            history['vali'].append({'step': step, 'drop_losses': scores['chromaticity'], 'main_vq': main_vq, 'vis_dirs': vis_dirs})
    if history['vali']:
        save_metas(outdir)
    return model, history


@torch.no_grad()
def render_views(model, dataset, outroot, relight_olat=False, relight_probes=False, opt_scale=None, writer=None, log=None,
                 num_p=None, p_i=None, **fast_render_kwargs):
    """The inference loops of the reference's test.py (:180-266: `raw_test` / `pd_test` / `pd_relit` passes): every view of
    `dataset` through `model.fast_render(mode='test', ...)`, its files queued into `outroot/batch{i:09d}` (i = the view's
    index in the sorted set) by the asynchronous `vis_batch`.  Rendering of view i + 1 overlaps the encoding of view i; the
    returned writer's `.flush()` waits for the files.  Views are independent: with `num_p` processes (default: the ranks of
    the process group) process `p_i` takes views p_i, p_i + num_p, ... -- multi-GPU batched inference with no collective.
    Returns (writer, number of views this process rendered)."""
    import os
    if num_p is None:
        num_p, p_i = parallel.world_size(), parallel.rank()
    order = {f: i for i, f in enumerate(sorted(dataset.files))}
    mine = {f for f, i in order.items() if i % num_p == p_i}
    n = 0
    for f in sorted(mine):
        batch = dataset.view(f)
        _, _, _, to_vis = model.fast_render(batch, mode='test', relight_olat=relight_olat, relight_probes=relight_probes,
                                            opt_scale=opt_scale, **fast_render_kwargs)
        writer = model.vis_batch(to_vis, os.path.join(outroot, 'batch{i:09d}'.format(i=order[f])), mode='test', writer=writer)
        n += 1
        if log is not None:
            log(f'view {order[f]} queued')
    if writer is None:
        from vqnerf_release_amd.decomp.nerfactor.util import vis
        writer = vis.default_writer()
    return writer, n


def fit_stage(config, outdir, dataset_train, dataset_vali=None, model=None, device='cuda', epochs=None, seed=None, log=print, graph=None):
    """Epoch loop of the stage-1 (`nfr_unit`) and stage-3 (`ref_nfr`) models: the shape_unit branch of trainvali.py:201-318.
    One max-colour-difference pair sample and one step per training view and epoch; `pretrain=True` with `bias_weight` for
    the first `pretrain_epochs` epochs; checkpoints every `ckpt_period`; every `vali_period` the summed loss terms
    (`loss.json`) and the validation views through `vis_batch` into `vis_vali/epoch{e:09d}/batch{b:09d}`, then `metas.json`.
    Returns (model, {'loss': [...], 'vali_dirs': [...]})."""
    import json, os
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    g = lambda k, cast, fb: cast(config.get('DEFAULT', k, fallback=str(fb)))
    data_type = config.get('DEFAULT', 'data_type')
    seed = g('random_seed', int, 0) if seed is None else seed
    torch.manual_seed(seed)
    rank0 = parallel.rank() == 0                                 # data parallel: own pairs per rank, files from rank 0
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 7919 * parallel.rank())
    os.makedirs(outdir, exist_ok=True)
    if model is None:
        model = get_model_class(config.get('DEFAULT', 'model'))(config)
        model.build_nets(device=device, seed=seed).to(device)
    _ = model.light
    model.register_trainable()
    # `graph`: the step captured once into a HIP graph and replayed (None = whenever eligible; under multi-rank RCCL the same explicit
    # opt-in as fit())
    use_graph = graph_default(device, model) if graph is None else bool(graph)
    if use_graph and parallel.world_size() > 1 and parallel.backend() != 'gloo' and not config.getboolean('DEFAULT', 'dp_graph', fallback=False) \
            and os.environ.get('VQN_DP_GRAPH', '0') in ('', '0'):
        use_graph = False
    opt, sched, clip = make_optimizer(config, model.trainable_variables, capturable=use_graph)
    ckptdir = os.path.join(outdir, 'checkpoints')
    latest, step = _latest_checkpoint(ckptdir), 0
    if latest is not None:
        state = torch.load(latest[1], map_location=device, weights_only=False)
        model.load_state_dict(state['net'])
        opt.load_state_dict(state['optimizer'])
        step = int(state['step'])
        log(f'Resumed from step {step}: {latest[1]}')
    trainer = Trainer(model, opt, clip=clip, sched=sched, graph=use_graph)
    epochs = config.getint('DEFAULT', 'epochs') if epochs is None else epochs
    pretrain_epochs = g('pretrain_epochs', int, 0)
    ckpt_period, vali_period = g('ckpt_period', int, 100), g('vali_period', int, 100)
    views_train = list(dataset_train.build_pipeline(no_shuffle=True))
    vali_views = list(dataset_vali.build_pipeline())[:g('vali_batches', int, 4)] if dataset_vali is not None and dataset_vali.get_n_views() else []
    import inspect
    accepted = set(inspect.signature(model.call).parameters)
    only = lambda d: {k: v for k, v in d.items() if k in accepted}      # nfr_unit takes `pretrain`, ref_nfr only `bias_weight`
    hist = {'loss': [], 'vali_dirs': []}
    for _ in range(step, epochs):
        kw = only({'pretrain': step < pretrain_epochs, 'bias_weight': 1.0})
        losses, sums = [], {}
        for view in views_train:
            batch = outer_sample(view, config, data_type, generator=gen, neighbour='max_diff')
            loss, _, loss_dict = trainer.train_iter(batch, dataset_train.bs * parallel.world_size(), **kw)
            losses.append(loss.detach().clone())
            for k, v in loss_dict.items():
                sums[k] = sums.get(k, 0.0) + v.detach().float().mean()
        pre = step < pretrain_epochs
        step += 1
        hist['loss'].append(float(torch.stack(losses).mean()))
        if step % ckpt_period == 0 and rank0:
            os.makedirs(ckptdir, exist_ok=True)
            torch.save({'step': step, 'net': model.state_dict(), 'optimizer': opt.state_dict()}, os.path.join(ckptdir, f'ckpt-{step}.pt'))
            log(f'Checkpointed step {step}: loss_train {hist["loss"][-1]:.6f}')
        if vali_views and vali_period > 0 and step % vali_period == 0 and rank0:
            edir = os.path.join(outdir, 'vis_vali', 'epoch{e:09d}'.format(e=step))
            os.makedirs(edir, exist_ok=True)
            with open(os.path.join(edir, 'loss.json'), 'w') as f:
                json.dump({k: float(v) for k, v in sums.items()}, f)
            writer = None
            with torch.no_grad():
                for b, view in enumerate(vali_views):
                    _, _, _, to_vis = model(view, mode='vali', **only({'pretrain': pre, 'bias_weight': 1.0}))
                    vdir = os.path.join(edir, 'batch{b:09d}'.format(b=b))
                    writer = model.vis_batch(to_vis, vdir, mode='vali')
                    hist['vali_dirs'].append(vdir)
            if writer is not None:
                writer.flush()
    if hist['vali_dirs']:
        save_metas(outdir)
    return model, hist
