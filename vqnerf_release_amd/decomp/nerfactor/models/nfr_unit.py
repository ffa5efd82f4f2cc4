"""Stage-1 continuous reflectance model: mirror of decomp/nerfvq_nfr3/nerfactor/models/nfr_unit.py
(nets :110-129, call :179-271, _render :273-306, light/gamma :309-330, heads :329-391, compute_loss :393-429).

`BrdfModel` holds what nfr_unit and vq_nfr share.  Two execution paths, selected per call:
  * no autograd graph needed (inference, `torch.no_grad()`): fused HIP kernels -- `vqn_mlp_chain_fwd` for the
    encoder / heads, `vqn_brdf_shade_fwd` for directions + microfacet BRDF + rendering-equation sum;
  * graph needed (training): the torch statements of the same arithmetic (networks/mlp.py, util/microfacet.py),
    on the GPU.  There is no CPU fallback: the fused path raises if the HIP library is missing.
"""
import os
import weakref

import vqnerf_release_amd

import numpy as np
import torch
import torch.nn as nn

from vqnerf_release_amd import _C
from vqnerf_release_amd.decomp import packing
from vqnerf_release_amd.decomp.nerfactor.models.shape import Model as ShapeModel
from vqnerf_release_amd.decomp.nerfactor.networks import mlp
from vqnerf_release_amd.decomp.nerfactor.util import img as imgutil, io as ioutil, math as mathutil, microfacet as micro_util


def fg_rows(alpha):
    """Row indices of the foreground rays (alpha > 0), ascending -- what `tf.where(mask)` / `tf.boolean_mask` select.  Taken
    ONCE per call and reused for every gather and scatter: a boolean-mask index costs a `nonzero` (a host sync) each time,
    twenty of them per rendered view.  Returns None when EVERY row is foreground (the index list is on the host after the
    `nonzero` anyway): `take_rows` / `scatter_rows` are then the identity and no gather or scatter is launched at all."""
    # The device-resident loaders hand the SAME alpha tensor to every call on a view (epochs, probes, validation): its index list is
    # remembered per tensor object (weakly: the entry dies with the tensor; an in-place write bumps `_version` and invalidates it),
    # so only the first call on a view pays the `nonzero` -- a host sync that otherwise leaves the GPU idle while the next call's
    # launches are prepared.
    # A writer torch does not see (a raw-pointer kernel through the C ABI, a DLPack peer) bumps no `_version`: such callers must
    # call `invalidate_fg_cache()` after rewriting an alpha buffer in place.
    key = id(alpha)
    sig = (alpha.data_ptr(), alpha._version, tuple(alpha.shape), alpha.device)
    hit = _FG_CACHE.get(key)
    if hit is not None and hit[0]() is alpha and hit[1] == sig:
        return hit[2]
    rows = (alpha[:, 0] > 0).nonzero(as_tuple=False).squeeze(1)
    rows = None if rows.numel() == alpha.shape[0] else rows
    if len(_FG_CACHE) > 4096:
        _FG_CACHE.clear()
    try:
        _FG_CACHE[key] = (weakref.ref(alpha, lambda _r, k=key: _FG_CACHE.pop(k, None)), sig, rows)
    except TypeError:
        pass
    return rows


def invalidate_fg_cache(alpha=None):
    """Forget the remembered foreground rows of `alpha` (all tensors when None).  For callers that rewrite an alpha buffer
    behind torch's back (ctypes kernels, DLPack peers): torch's `_version` counter does not see such writes."""
    if alpha is None:
        _FG_CACHE.clear()
    else:
        _FG_CACHE.pop(id(alpha), None)


_FG_CACHE = {}


def scatter_rows(mask, x, n):
    """tf.scatter_nd(tf.where(mask), x, (n, c)): rows of x back to their ray slots, zeros elsewhere.  mask None = every
    row is foreground (see `take_rows`)."""
    if mask is None:
        return x
    out = x.new_zeros((n,) + tuple(x.shape[1:]))
    out[mask] = x
    return out


def take_rows(mask, *tensors):
    """tf.boolean_mask over dim 0 for each tensor (None passes through).  mask None = keep everything: a boolean gather has
    a data-dependent shape (a host sync), which a captured training step cannot have."""
    out = tuple(t if (t is None or mask is None) else t[mask] for t in tensors)
    return out if len(out) > 1 else out[0]


class LazyRows:
    """`full[rows]` not yet gathered: the visibility buffer of a view is 2 KB per pixel, and the fused shading kernel can read the
    foreground rows in place (`vqn_brdf_shade_fwd_rows`), so the tf.boolean_mask copy of vq_nfr.py:558-559 is only made for the
    paths that need a dense tensor (`dense()`)."""

    def __init__(self, full, rows):
        self.full, self.rows = full, rows
        self._dense = None

    def dense(self):
        if self._dense is None:
            self._dense = self.full[self.rows]
        return self._dense

    @property
    def shape(self):
        return (self.rows.numel(),) + tuple(self.full.shape[1:])


def dense_rows(t):
    return t.dense() if isinstance(t, LazyRows) else t


class ShadeFunction(torch.autograd.Function):
    """Directions + microfacet BRDF + rendering-equation sum as one differentiable op (vqn_brdf_shade_fwd(raw) /
    vqn_brdf_shade_bwd).  Differentiable inputs: light [L,3] and, per material set, albedo [N,3], spec [N,3], rough [N,1];
    outputs: the plain sums over lights per set (gamma / clip are applied by the caller in torch) and the corrected normal."""

    @staticmethod
    def forward(ctx, geom, light, *mats):
        xyz, normal, rayo, lvis, lxyz, lareas = geom[:6]
        clip = len(geom) > 6 and bool(geom[6])       # the sums through clip_by_value_preserve_gradient(0, 1) inside the kernel (identity gradient: backward unchanged)
        light = light.detach().float().reshape(-1, 3).contiguous()
        sets = [tuple(t.detach().float().contiguous() for t in mats[3 * i:3 * i + 3]) for i in range(len(mats) // 3)]
        sets = [(a, s.expand(-1, 3).contiguous(), r) for a, s, r in sets]
        out = _C.brdf_shade_fwd(xyz, normal, rayo, lvis, lxyz, lareas, light, sets, want_normal=True, raw=2 if clip else 1)
        ctx.geom, ctx.light, ctx.sets = geom[:6], light, sets
        ctx.spec_widths = [mats[3 * i + 1].shape[1] for i in range(len(sets))]
        ctx.mark_non_differentiable(out['normal'])
        return (out['normal'],) + tuple(out['rgb'])

    @staticmethod
    def backward(ctx, _g_normal, *g_sums):
        xyz, normal, rayo, lvis, lxyz, lareas = ctx.geom
        gs = [torch.zeros_like(ctx.sets[i][0]) if g is None else g.contiguous() for i, g in enumerate(g_sums)]
        grads, g_light = _C.brdf_shade_bwd(xyz, normal, rayo, lvis, lxyz, lareas, ctx.light, ctx.sets, gs)
        flat = []
        for (ga, gsp, gr), w in zip(grads, ctx.spec_widths):
            flat += [ga, gsp if w == 3 else gsp.sum(-1, keepdim=True), gr]
        return (None, g_light) + tuple(flat)


class _KsSplit(torch.autograd.Function):
    """(spec, albedo) = (ks basecolor, (1 - ks) basecolor) (vq_nfr.py:590-592, nfr_unit.py:215-217): one launch forward, one backward
    (vqn_ks_split_fwd / _bwd) instead of three + six framework launches; the same roundings as the torch statement."""

    @staticmethod
    def forward(ctx, ks, basecolor):
        k, b = ks.detach().float().contiguous(), basecolor.detach().float().contiguous()
        ctx.save_for_backward(k, b)
        albedo, spec = _C.ks_split_fwd(b, k)
        return spec, albedo

    @staticmethod
    def backward(ctx, g_spec, g_albedo):
        k, b = ctx.saved_tensors
        c = lambda t: None if t is None else t.float().contiguous()
        g_bc, g_ks = _C.ks_split_bwd(b, k, c(g_albedo), c(g_spec))
        return g_ks, g_bc


def ks_split(ks, basecolor):
    """-> (spec, albedo)"""
    if ks.is_cuda and ks.dtype == torch.float32 and basecolor.dtype == torch.float32 and basecolor.shape[1] == 3 and ks.shape[1] in (1, 3):
        return _KsSplit.apply(ks, basecolor)
    return ks * basecolor, (1 - ks) * basecolor


class _PackCache:
    def __init__(self):
        self.key, self.value = None, None

    def get(self, params, build):
        # besides the tensors' `_version`s the process-wide weights epoch: a fused / capturable Adam step and a replayed HIP graph
        # move the weights without bumping any version (vqnerf_release_amd/__init__.py)
        # (FROZEN parameters -- the stage-2 parts inside a stage-3 training step -- are moved by no optimiser and no graph replay: their packs
        #  follow the tensors' versions only, and are NOT rebuilt after every step of the nets that do train; a rebuild reads small biases
        #  back to the host, which a captured step cannot record)
        epoch = vqnerf_release_amd.weights_epoch() if any(p.requires_grad for p in params) else -1
        key = (epoch,) + tuple((id(p), p._version, str(p.device)) for p in params)
        if key != self.key:
            with torch.no_grad():
                self.value = build()
            self.key = key
        return self.value


class BrdfModel(ShapeModel):
    HEADS = ('diff', 'spec', 'rough')

    def __init__(self, config, debug=False):
        self.data_type = config.get('DEFAULT', 'data_type')
        self.z_dim = config.getint('DEFAULT', 'conv_width', fallback=256)
        super().__init__(config, debug=debug)
        self._light = None
        self._gamma_index, self._gamma_bias = None, None
        self._plans, self._packs, self._engines = {}, {}, {}
        self.matrix_mode = 'f32'         # 'f16s': inference MLP stacks on the split-precision (f16 hi/lo MFMA) kernel, ~1e-6 relative;
                                         # 'x3' (round 4): on the exact-split stack kernel (bf16 piece triples, f32-level products)
        self.assume_foreground = False   # True: callers promise alpha > 0 everywhere (vq_nfr.Model.call skips the boolean gathers)
        self.train_backend = 'hip'       # 'hip': fused shading fwd/bwd kernels under autograd; 'torch': torch statements
        # the reference's tf.debugging.check_numerics guards (vq_nfr.py:731, :783, :802, :815, :827, :985 and their twins in
        # nfr_unit.py / ref_nfr.py): optional here -- each is a host sync -- on with `debug=True` or VQN_CHECK_NUMERICS=1
        self.check_numerics = bool(debug) or os.environ.get('VQN_CHECK_NUMERICS', '0') not in ('', '0')
        self._novel_lights()

    def weights_changed(self):
        """Tell the inference-side caches (weight packs, codebook fragments) that parameters were rewritten in a way torch's
        `_version` counters do not record: a replayed HIP graph (`Trainer(graph=True)`), a raw-pointer writer through the C ABI."""
        vqnerf_release_amd.weights_changed()

    def _apply(self, fn, *args, **kwargs):
        """.to(device) / .float() also move the relighting maps (plain tensors in dicts, not buffers)."""
        out = super()._apply(fn, *args, **kwargs)
        for name in ('novel_probes', 'novel_olat'):
            d = getattr(self, name, None)
            if d:
                for k in list(d):
                    d[k] = fn(d[k])
        return out

    def _load_light(self, path, resize=False):
        """.hdr / .npy probe -> float32 [h, w, 3] (nfr_unit.py / vq_nfr.py `_load_light`; .exr needs OpenEXR, absent here)."""
        ext = os.path.basename(path).split('.')[-1]
        if ext == 'hdr':
            arr = ioutil.read_hdr(path)
        elif ext == 'npy':
            arr = np.load(path)
        else:
            raise NotImplementedError(ext)
        t = torch.tensor(np.asarray(arr), dtype=torch.float32)
        if resize and t.shape[0] != self.light_res[0]:
            h, w = self.light_res[0], int(t.shape[1] / t.shape[0] * self.light_res[0])
            t = torch.nn.functional.interpolate(t.permute(2, 0, 1)[None], size=(h, w), mode='bilinear', align_corners=False,
                                                antialias=True)[0].permute(1, 2, 0).contiguous()
        return t

    def _novel_lights(self):
        """Relighting conditions of test time (nfr_unit.py:61-91): four one-light-at-a-time maps (row 4, columns 0 / 8 / 16 /
        24, `olat_inten` over an `ambient_inten` floor when the background is white) and every .hdr / .npy probe under
        `test_envmap_dir` (sorted by name), at the resolution of the light grid."""
        from collections import OrderedDict
        from glob import glob
        cfg = self.config
        olat_inten = cfg.getfloat('DEFAULT', 'olat_inten', fallback=200)
        ambi = cfg.getfloat('DEFAULT', 'ambient_inten', fallback=0) if self.white_bg else 0.0
        dev = self.lxyz.device
        self.novel_olat = OrderedDict()
        for i in [4]:
            for j in [0, 8, 16, 24]:
                if i < self.light_res[0] and j < self.light_res[1]:
                    env = torch.full(self.light_res + (3,), float(ambi))
                    env[i, j, :] += olat_inten
                    self.novel_olat['%04d-%04d' % (i, j)] = env.to(dev)
        self.novel_probes = OrderedDict()
        d = cfg.get('DEFAULT', 'test_envmap_dir', fallback='')
        if d and os.path.isdir(d):
            for path in sorted(glob(os.path.join(d, '*.hdr')) + glob(os.path.join(d, '*.npy'))):
                self.novel_probes[os.path.basename(path)[:-4]] = self._load_light(path, resize=True).to(dev)

    # ------------------------------------------------------------------ parameters
    def _standard_nets(self, suffix):
        z = self.z_dim
        head = lambda out: mlp.Network([z, z // 2, out], act=['relu'] * 2 + ['sigmoid'], skip_at=[1])
        return {'diff_' + suffix: head(3), 'spec_' + suffix: head(1), 'rough_' + suffix: head(1)}

    def _encoder_nets(self):
        w = self.config.getint('DEFAULT', 'mlp_width')
        z = self.z_dim
        return {'fine_enc': mlp.Network([w] * 4, act=['relu'] * 4, skip_at=[2]),
                'bottleneck': mlp.Network([w, z, z], act=[None, 'relu', 'sigmoid'])}

    def build_nets(self, device=None, seed=None):
        """Create every Dense kernel (Keras would do so lazily at first call)."""
        gen = None
        if seed is not None:
            gen = torch.Generator().manual_seed(int(seed))
        d_xyz = getattr(self.embedder['xyz'], 'out_dims', 3)
        for name, net in self.net.items():
            d_in = d_xyz if name == 'fine_enc' else (self.config.getint('DEFAULT', 'mlp_width') if name == 'bottleneck' else self.z_dim)
            net.build(d_in, device=device, generator=gen)
        self.register_trainable()
        return self

    @property
    def light(self):
        if self._light is None:
            path = self.config.get('DEFAULT', 'light_path', fallback='')
            if path and os.path.exists(path):
                arr = torch.tensor(np.load(path), dtype=torch.float32)
            else:
                inten = self.config.getfloat('DEFAULT', 'light_init_val', fallback=0.5)
                arr = torch.ones(self.light_res + (3,)) * inten
            self._light = nn.Parameter(arr.to(self.lxyz.device))
        return mathutil.clip_preserve_gradient(self._light, 0.0, float('inf'))

    def set_light(self, arr):
        self._light = nn.Parameter(torch.as_tensor(arr, dtype=torch.float32).to(self.lxyz.device))

    @property
    def gamma(self):
        if self._gamma_index is None:
            self._gamma_index = nn.Parameter(torch.ones(1, device=self.lxyz.device))
            self._gamma_bias = nn.Parameter(torch.ones(1, device=self.lxyz.device))
        return torch.cat([self._gamma_bias, mathutil.clip_preserve_gradient(self._gamma_index, 0.0, 5.0)], 0)

    # ------------------------------------------------------------------ dispatch
    def _needs_graph(self, *tensors):
        if not torch.is_grad_enabled():
            return False
        return any(p.requires_grad for p in self.parameters()) or any(t is not None and t.requires_grad for t in tensors)

    def _fused(self, *tensors):
        """True when this call runs on the fused HIP kernels (no graph needed).  CPU tensors then raise: the
        torch statements below are the autograd path, not a fallback."""
        if self._needs_graph(*tensors):
            return False
        _C.require_device(tensors[0], type(self).__name__)
        if not getattr(self.embedder['xyz'], 'fused_ok', lambda: False)():
            raise _C.VqnError('the fused encoder needs the standard positional encoding (pos_enc = True)')
        return True

    # ------------------------------------------------------------------ fused layer programs
    def _enc_program(self):
        pkey = 'enc:' + self._chain_mode()
        if pkey not in self._plans:
            emb = self.embedder['xyz']
            b = packing.ChainBuilder('posenc', emb.out_dims, n_freqs=emb.n_freqs, mode=self._chain_mode())
            fe, bn = self.net['fine_enc'], self.net['bottleneck']
            y = b.mlp('fine_enc', fe.widths, fe.act, fe.skip_at, b.input)
            assert not isinstance(y, list)
            b.mlp('bottleneck', bn.widths, bn.act, bn.skip_at, y, out_slot=0, small_last=False)
            self._plans[pkey] = b.build()
        return self._plans[pkey]

    def _head_program(self, names, in_dim=None):
        in_dim = in_dim or self.z_dim
        key = 'heads:%s:%d:' % (self._chain_mode(), in_dim) + ','.join(names)
        if key not in self._plans:
            b = packing.ChainBuilder('raw', in_dim, mode=self._chain_mode())
            x = b.input
            resident = self._chain_mode() == 'f32' and all(self._is_std_head(self.net[n]) and self.net[n].widths[1] <= 128 for n in names)
            for slot, name in enumerate(names):
                net = self.net[name]
                if resident:
                    # [w0, w1, c] with the input concatenated into the last layer.  The input image (32 rows at z_dim = 256) stays in
                    # LDS for the whole program: layer 1's output is written IN PLACE over layer 0's (its own K operand; the kernel
                    # holds the accumulators across a barrier), so input + one wide activation = 64 rows = two workgroups per CU and
                    # ONE fetch of the input per point tile instead of two per head.  Same arithmetic and order as the form below.
                    y0 = b.dense(f'{name}/0', [x], net.widths[0], net.act[0], keep=[x])
                    y1 = b.dense(f'{name}/1', [y0], net.widths[1], net.act[1], keep=[x], over=y0)
                    b.dense_small(f'{name}/2', [y1, x], net.widths[2], net.act[2], slot)
                elif self._is_std_head(net):
                    # the input image is NOT kept in LDS while the two wide activations are live -- it is fetched again (L2) for
                    # the last layer, and for the next head.  That keeps the program at 64 rows, i.e. two workgroups per CU.
                    if x is None:
                        x = b.reload_input()
                    y0 = b.dense(f'{name}/0', [x], net.widths[0], net.act[0])
                    y1 = b.dense(f'{name}/1', [y0], net.widths[1], net.act[1])
                    x = b.reload_input(keep=[y1])
                    b.dense_small(f'{name}/2', [y1, x], net.widths[2], net.act[2], slot)
                else:
                    if x is None:
                        x = b.reload_input()
                    b.mlp(name, net.widths, net.act, net.skip_at, x, keep=[x], out_slot=slot)
            self._plans[key] = b.build()
        return self._plans[key]

    @staticmethod
    def _is_std_head(net):
        return len(net.widths) == 3 and net.skip_at == [1] and net.widths[2] <= 4

    def _program_pack(self, key, plan, nets):
        cache = self._packs.setdefault(key, _PackCache())
        params, pdict = [], {}
        for name in nets:
            for i, layer in enumerate(self.net[name].layers):
                params += [layer.kernel, layer.bias]
                pdict[f'{name}/{i}'] = (layer.kernel.detach(), layer.bias.detach())
        return cache.get(params, lambda: plan.pack(pdict))

    def _enc_heads_program(self, names):
        """Encoder + the `names` head family in ONE program (f32 kernels, standard heads): z stays in LDS as the heads' input --
        it is written to HBM once (slot 0, the VQ branch reads it) and never read back for the continuous branch.  Same per-point
        arithmetic and order as the two separate programs (bit-identical outputs, tests/test_gpu_decomp.py); 16 layers, 64 rows."""
        key = 'enc+heads:' + ','.join(names)
        if key not in self._plans:
            emb = self.embedder['xyz']
            b = packing.ChainBuilder('posenc', emb.out_dims, n_freqs=emb.n_freqs, mode='f32')
            fe, bn = self.net['fine_enc'], self.net['bottleneck']
            y = b.mlp('fine_enc', fe.widths, fe.act, fe.skip_at, b.input)
            z = b.mlp('bottleneck', bn.widths, bn.act, bn.skip_at, y, out_slot=0, small_last=False)
            for slot, name in enumerate(names):
                net = self.net[name]
                y0 = b.dense(f'{name}/0', [z], net.widths[0], net.act[0], keep=[z])
                y1 = b.dense(f'{name}/1', [y0], net.widths[1], net.act[1], keep=[z], over=y0)
                b.dense_small(f'{name}/2', [y1, z], net.widths[2], net.act[2], slot + 1)
            self._plans[key] = b.build()
        return self._plans[key]

    def _can_fuse_enc_heads(self, names):
        fe, bn = self.net['fine_enc'], self.net['bottleneck']
        n_layers = len(fe.widths) + len(bn.widths) + 3 * len(names)
        return (self.matrix_mode == 'f32' and len(names) <= 3 and n_layers <= packing.MAX_LAYERS and bn.widths[-1] == self.z_dim
                and all(self._is_std_head(self.net[n]) and self.net[n].widths[1] <= 128 for n in names))

    def _fused_enc_heads(self, pts, names):
        """-> (z [N, z_dim], head outputs...) in one launch."""
        plan = self._enc_heads_program(names)
        wbuf, desc = self._program_pack('enc+heads:' + ','.join(names), plan, ['fine_enc', 'bottleneck'] + list(names))
        widths = [self.z_dim] + [self.net[n].widths[-1] for n in names]
        outs = _C.mlp_chain_fwd(desc, wbuf, pts.detach().float().contiguous(), widths)
        return outs[0], outs[1:]

    def enc_and_heads(self, pts, suffix):
        """(z, basecolor | albedo, ks | spec, rough) of `pts`: `_pred_bias_at` followed by `_all_heads(z, suffix)`, as one fused
        program where the path allows it (no graph, f32 kernels, standard heads)."""
        names = [h + '_' + suffix for h in self.HEADS]
        if self.matrix_mode == 'x3' and self._fused(pts):
            res = self._x3_infer(names, pts, True)
            if res is not None:
                z, (d, s_, r) = res
                return (self._numerics(z, 'Z'), self._numerics(self._albedo_affine(d), 'Albedo'), self._numerics(s_, 'Specular'),
                        self._numerics(r, 'Roughness'))
        if self._fused(pts) and self._can_fuse_enc_heads(names) and self._plan_fits_two_workgroups(names):
            z, (d, s_, r) = self._fused_enc_heads(pts, names)
            return (self._numerics(z, 'Z'), self._numerics(self._albedo_affine(d), 'Albedo'), self._numerics(s_, 'Specular'),
                    self._numerics(r, 'Roughness'))
        eng = self._stack_engine(names, pts, with_encoder=True)
        if eng is not None:                                   # training: encoder + the three heads, one launch each way (round 4)
            from vqnerf_release_amd.decomp.refl_train import ReflStackFunction
            z, d, s_, r = ReflStackFunction.apply(eng, pts, *eng.params())
            return (self._numerics(z, 'Z'), self._numerics(self._albedo_affine(d), 'Albedo'), self._numerics(s_, 'Specular'),
                    self._numerics(r, 'Roughness'))
        z = self._pred_bias_at(pts)
        return (z,) + tuple(self._all_heads(z, suffix))

    def _plan_fits_two_workgroups(self, names):
        return self._enc_heads_program(names).n_waves == 4

    def _fused_enc(self, pts):
        if self.matrix_mode == 'x3':
            res = self._x3_infer([], pts, True)
            if res is not None:
                return res[0]
        plan = self._enc_program()
        wbuf, desc = self._program_pack('enc:' + self._chain_mode(), plan, ['fine_enc', 'bottleneck'])
        return _C.mlp_chain_fwd(desc, wbuf, pts.detach().float().contiguous(), [self.z_dim], mode=self._chain_mode())[0]

    def _fused_heads(self, z, names):
        if self.matrix_mode == 'x3':
            res = self._x3_infer(list(names), z, False)
            if res is not None:
                return res[1]
        plan = self._head_program(names, z.shape[1])
        wbuf, desc = self._program_pack('heads:%s:%d:' % (self._chain_mode(), z.shape[1]) + ','.join(names), plan, names)
        widths = [self.net[n].widths[-1] for n in names]
        return _C.mlp_chain_fwd(desc, wbuf, z.detach().float().contiguous(), widths, mode=self._chain_mode())

    # ------------------------------------------------------------------ training engines (tile programs)
    def _train_hip(self, x):
        return self.train_backend == 'hip' and x.is_cuda and getattr(self.embedder['xyz'], 'fused_ok', lambda: False)()

    # 'x3': the dedicated exact-split kernels of csrc/refl_train_x3.hip (decomp/refl_train.py) wherever the stack has their shape;
    # 'prog': the interpreted tile programs of rounds 1-3 (decomp/train_programs.py).  VQN_REFL_TRAIN overrides.
    REFL_TRAIN_DEFAULT = 'x3'

    def _stack_engine(self, names, x, with_encoder):
        """The dedicated training engine for (encoder +) the `names` heads, or None when this call is not a HIP training call or the
        stack is outside what the kernels run (then the interpreted programs / torch statements take it)."""
        if not (self._train_hip(x) and not self._fused(x) and os.environ.get('VQN_REFL_TRAIN', self.REFL_TRAIN_DEFAULT) == 'x3'):
            return None
        key = ('stack', with_encoder, str(x.device)) + tuple(names)
        if key not in self._engines:
            from vqnerf_release_amd.decomp.refl_train import ReflStackEngine
            enc = [self.net['fine_enc'], self.net['bottleneck']] if with_encoder else None
            heads = [self.net[n] for n in names]
            emb = self.embedder['xyz']
            ok = ReflStackEngine.supports(enc, heads, self.z_dim, getattr(emb, 'out_dims', 0) if with_encoder else 0)
            self._engines[key] = ReflStackEngine(enc, emb.n_freqs if with_encoder else 0, heads, self.z_dim, x.device) if ok else None
        eng = self._engines[key]
        if eng is None or (not with_encoder and x.shape[1] != self.z_dim):
            return None
        if with_encoder and x.requires_grad:     # ReflStackFunction has no adjoint for the points: the interpreted programs / torch take it
            return None
        return eng if any(p.requires_grad for p in eng.params()) or x.requires_grad else None

    # ---- matrix_mode = 'x3' (round 4, opt-in like 'f16s'): inference on the exact-split stack kernel of the trainers
    # (csrc/refl_train_x3.hip with nothing kept for a backward) -- f32-level products (six bf16 MFMAs each), no operand-range caveat
    def _x3_infer(self, names, x, with_encoder):
        """(z rows | None, head outputs) on the exact-split engine, or None when the stack is outside its shape (then the f32 chain)."""
        key = ('stack', with_encoder, str(x.device)) + tuple(names)
        if key not in self._engines:
            from vqnerf_release_amd.decomp.refl_train import ReflStackEngine
            enc = [self.net['fine_enc'], self.net['bottleneck']] if with_encoder else None
            heads = [self.net[n] for n in names]
            emb = self.embedder['xyz']
            ok = ReflStackEngine.supports(enc, heads, self.z_dim, getattr(emb, 'out_dims', 0) if with_encoder else 0)
            self._engines[key] = ReflStackEngine(enc, emb.n_freqs if with_encoder else 0, heads, self.z_dim, x.device) if ok else None
        eng = self._engines[key]
        if eng is None or (not with_encoder and x.shape[1] != self.z_dim):
            return None
        if with_encoder and x.requires_grad:     # ReflStackFunction has no adjoint for the points: the interpreted programs / torch take it
            return None
        cache = self._packs.setdefault(('x3',) + key, _PackCache())
        packs = cache.get(eng.params(), lambda: eng.build_packs(eng.params()))
        return eng.infer(x, packs)

    def _chain_mode(self):
        """mode of the descriptor-driven chain kernels: they have no exact-split variant ('x3' stacks outside the stack kernel's shape
        run the f32 chain)"""
        return 'f32' if self.matrix_mode == 'x3' else self.matrix_mode

    def _enc_engine(self, device):
        if 'enc' not in self._engines:
            from vqnerf_release_amd.decomp.train_programs import EncoderEngine
            self._engines['enc'] = EncoderEngine(self.net['fine_enc'], self.net['bottleneck'], self.embedder['xyz'].n_freqs, device)
        return self._engines['enc']

    def _heads_engine(self, names, device):
        key = 'heads:' + ','.join(names)
        if key not in self._engines:
            from vqnerf_release_amd.decomp.train_programs import HeadsEngine
            self._engines[key] = HeadsEngine([self.net[n] for n in names], self.z_dim, device)
        return self._engines[key]

    # ------------------------------------------------------------------ reference-named pieces
    def _numerics(self, x, message):
        return mathutil.check_numerics(x, message) if self.check_numerics else x

    _HEAD_MESSAGES = {'diff': 'Albedo', 'spec': 'Specular', 'rough': 'Roughness'}

    def _pred_bias_at(self, pts):
        """xyz [N,3] -> z [N,z_dim]  (nfr_unit.py:329-342; vq_nfr.py:771-784 is the same function)."""
        if self._fused(pts):
            return self._numerics(self._fused_enc(pts), 'Z')
        if self._train_hip(pts):
            from vqnerf_release_amd.decomp.train_programs import EncoderFunction
            layers = list(self.net['fine_enc'].layers) + list(self.net['bottleneck'].layers)
            return self._numerics(EncoderFunction.apply(self._enc_engine(pts.device), pts, *[l.kernel for l in layers],
                                                        *[l.bias for l in layers]), 'Z')
        return self._numerics(self.net['bottleneck'](self.net['fine_enc'](self.embedder['xyz'](pts))), 'Z')

    def _head(self, name, z):
        msg = self._HEAD_MESSAGES.get(name.split('_')[0], name)
        if self._fused(z):
            return self._numerics(self._fused_heads(z, [name])[0], msg)
        if self._train_hip(z) and z.shape[1] == self.z_dim and any(p.requires_grad for p in self.net[name].parameters()):
            from vqnerf_release_amd.decomp.train_programs import HeadsFunction
            net = self.net[name]
            return self._numerics(HeadsFunction.apply(self._heads_engine([name], z.device), z, *[l.kernel for l in net.layers],
                                                      *[l.bias for l in net.layers])[0], msg)
        return self._numerics(self.net[name](z), msg)

    def _albedo_affine(self, albedo):
        slope = self.config.getfloat('DEFAULT', 'albedo_slope', fallback=1.0)
        bias = self.config.getfloat('DEFAULT', 'albedo_bias', fallback=0.0)
        return albedo if (slope == 1.0 and bias == 0.0) else slope * albedo + bias

    def _pred_diff_at(self, z):
        return self._albedo_affine(self._head('diff_out', z))

    def _pred_spec_at(self, z):
        return self._head('spec_out', z)

    def _pred_rough_at(self, z):
        return self._head('rough_out', z)

    def _all_heads(self, z, suffix, keep_input=False):
        """(basecolor|albedo [N,3], ks|spec [N,1|3], rough [N,1]) of the `suffix` head family in one launch.  keep_input: a fourth value,
        the rows `z` as a further consumer should read them (the same values; on the training kernels an output of the heads' node, so
        that the consumer's adjoint is added inside the backward kernel -- ReflStackKeepFunction)."""
        names = [h + '_' + suffix for h in self.HEADS]
        z_keep = z
        if self._fused(z):
            d, s, r = self._fused_heads(z, names)
        elif self._stack_engine(names, z, with_encoder=False) is not None:
            from vqnerf_release_amd.decomp.refl_train import ReflStackFunction, ReflStackKeepFunction
            eng = self._stack_engine(names, z, with_encoder=False)
            if keep_input and z.requires_grad and z.dtype == torch.float32 and z.is_contiguous():
                z_keep, d, s, r = ReflStackKeepFunction.apply(eng, z, *eng.params())
            else:
                d, s, r = ReflStackFunction.apply(eng, z, *eng.params())
        elif self._train_hip(z):
            from vqnerf_release_amd.decomp.train_programs import HeadsFunction
            nets = [self.net[n] for n in names]
            d, s, r = HeadsFunction.apply(self._heads_engine(names, z.device), z, *[l.kernel for n in nets for l in n.layers],
                                          *[l.bias for n in nets for l in n.layers])
        else:
            d, s, r = (self.net[n](z) for n in names)
        out = (self._numerics(self._albedo_affine(d), 'Albedo'), self._numerics(s, 'Specular'), self._numerics(r, 'Roughness'))
        return out + (z_keep,) if keep_input else out

    @staticmethod
    def _normal_correct(normal, surf2c):
        cos = (normal * surf2c).sum(-1, keepdim=True)
        return torch.where(cos >= 0, normal, -normal)

    def _eval_brdf_at(self, pts2l, pts2c, normal, albedo, spec, rough, chunk_size=None):
        return micro_util.get_brdf(pts2l, pts2c, normal, albedo=albedo, rough=rough, f0=spec)

    def _integrate(self, brdf, lvis_eff, cos, light):
        contrib = brdf * (lvis_eff[:, :, None] * light.reshape(1, -1, 3)) * cos[:, :, None] * self.lareas.reshape(1, -1, 1)
        rgb = contrib.sum(1)
        if self.data_type != 'nerf':
            g = self.gamma
            rgb = (rgb * g[0]) ** g[1]
        return mathutil.clip_preserve_gradient(rgb, 0.0, 1.0)

    def _render(self, brdf, l, n, light_vis=None, relight_olat=False, relight_probes=False, dst_env=None,
                white_light_override=False, light=None):
        """torch statement of the rendering-equation sum (nfr_unit.py:273-306, vq_nfr.py:694-733)."""
        light_vis = dense_rows(light_vis)
        if light is None:
            light = self.light if dst_env is None else self.novel_probes[dst_env]
        if white_light_override:
            light = torch.ones_like(light)
        cos = torch.einsum('ijk,ik->ij', l, n)
        front = (cos > 0).to(cos.dtype)
        vis = front if light_vis is None else front * light_vis
        rgb = self._integrate(brdf, vis, cos, light)
        rgb_probes = None
        if relight_probes:                                       # True: the loaded probes; a list: those maps (OLAT and / or probes)
            maps = list(self.novel_probes.values()) if relight_probes is True else list(relight_probes)
            rgb_probes = self._numerics(torch.stack([self._integrate(brdf, vis, cos, lp) for lp in maps], 1), 'Light Probe Renders')
        return rgb, None, rgb_probes

    def fg_lvis(self, lvis, mask, like):
        """The foreground rows of the visibility buffer: gathered lazily on the device inference path (see LazyRows)."""
        if lvis is None or mask is None:
            return lvis
        if like.is_cuda and not self._needs_graph(like) and lvis.dtype == torch.float32 and lvis.is_contiguous():
            return LazyRows(lvis, mask)
        return take_rows(mask, lvis)

    def _shade(self, xyz, normal, rayo, lvis, materials, split=False, light=None, probes=None):
        """Fused directions + BRDF + integral for 1-2 material sets (no graph).  Returns the dict of _C.brdf_shade_fwd."""
        light = self.light if light is None else light
        gamma = None if self.data_type == 'nerf' else self.gamma.detach()
        mats = [(a.detach().float().contiguous(), s.detach().float().expand(-1, 3).contiguous(),
                 r.detach().float().contiguous()) for a, s, r in materials]
        lvis_rows = None
        if isinstance(lvis, LazyRows):
            lvis, lvis_rows = lvis.full, lvis.rows.contiguous()
        out = _C.brdf_shade_fwd(xyz.detach().float().contiguous(), normal.detach().float().contiguous(),
                                rayo.detach().float().contiguous(),
                                None if lvis is None else lvis.detach().float().contiguous(),
                                self.lxyz.reshape(-1, 3).contiguous(), self.lareas.reshape(-1).contiguous(),
                                light.detach().float().reshape(-1, 3).contiguous(), mats, gamma=gamma,
                                want_normal=True, want_split=split, probes=probes, lvis_rows=lvis_rows)
        self._numerics(out.get('rgb_probes'), 'Light Probe Renders')
        return out

    def _shade_train(self, xyz, normal, rayo, lvis, materials, light=None):
        """Shading with gradients (albedo / spec / rough / light) through the fused forward + backward kernels."""
        light = self.light if light is None else light
        lvis = dense_rows(lvis)
        c = lambda t: t.detach().float().contiguous()
        # data_type 'nerf' has no gamma curve: the [0, 1] clip with identity gradient is applied by the kernel itself (raw = 2, the
        # arithmetic of vqn_clip_preserve) -- it was a launch per material set
        in_kernel_clip = self.data_type == 'nerf'
        geom = (c(xyz), c(normal), c(rayo), None if lvis is None else c(lvis), self.lxyz.reshape(-1, 3).contiguous(),
                self.lareas.reshape(-1).contiguous(), in_kernel_clip)
        flat = [t for m in materials for t in m]
        res = ShadeFunction.apply(geom, light.reshape(-1, 3), *flat)
        n_pred, sums = res[0], res[1:]
        if in_kernel_clip:
            return {'rgb': list(sums), 'normal': n_pred, 'rgb_diff': None, 'rgb_spec': None}
        rgb = []
        for sm in sums:
            g = self.gamma
            sm = (sm * g[0]) ** g[1]
            rgb.append(mathutil.clip_preserve_gradient(sm, 0.0, 1.0))
        return {'rgb': rgb, 'normal': n_pred, 'rgb_diff': None, 'rgb_spec': None}

    def _unpack(self, batch):
        if self.data_type == 'nerf':
            id_, hw, rayo, rayd, rgb, alpha, pred_alpha, xyz, normal, lvis = batch
        else:
            id_, hw, rayo, rayd, rgb, alpha, pred_alpha, xyz, normal = batch
            lvis = None
        return id_, hw, rayo, rayd, rgb, alpha, pred_alpha, xyz, normal, lvis

    def vis_batch(self, data_dict, outdir, mode='train', dump_raw_to=None, light_vis_h=256, alpha_thres=0.8, simp=False,
                  full_vis_path=None, writer=None):
        """Per-view output files of vq_nfr.py:988-1134, queued on an asynchronous writer (util/vis.py); returns the writer
        (`.flush()` joins)."""
        from vqnerf_release_amd.decomp.nerfactor.util import vis
        return vis.vis_batch(self, data_dict, outdir, mode=mode, light_vis_h=light_vis_h, alpha_thres=alpha_thres, simp=simp,
                             full_vis_path=full_vis_path, writer=writer)


class Model(BrdfModel):
    def _init_net(self):
        net = self._standard_nets('out')
        net.update(self._encoder_nets())
        return net

    def call(self, batch, mode='train', pretrain=False, relight_olat=False, relight_probes=False, save_z=False,
             opt_scale=None, bias_weight=None):
        self._validate_mode(mode)
        id_, hw, rayo, rayd, rgb, alpha, pred_alpha, xyz, normal, lvis = self._unpack(batch)
        gt = {'rgb': rgb, 'normal': normal, 'alpha': alpha, 'xyz': xyz}
        # (`assume_foreground`, set by a capturing Trainer: training batches of outer_sample hold foreground rows only -- no boolean gather,
        #  no host sync; validation views do have background rows)
        mask = None if (self.assume_foreground and mode == 'train') else fg_rows(alpha)
        n = alpha.shape[0]
        rayo, rgb_m, xyz_m, normal_m = take_rows(mask, rayo), take_rows(mask, rgb), take_rows(mask, xyz), take_rows(mask, normal)
        lvis_m = self.fg_lvis(lvis, mask, xyz_m)
        z_bias, basecolor, ks, rough = self.enc_and_heads(xyz_m, 'out')
        spec, albedo = ks_split(ks, basecolor)
        if not self._fused(xyz_m, albedo, spec, rough) and self.train_backend == 'hip' and xyz_m.is_cuda and mode == 'train':
            sh = self._shade_train(xyz_m, normal_m, rayo, lvis_m, [(albedo, spec, rough)])
            rgb_pred, normal_pred = sh['rgb'][0], sh['normal']
        elif not self._fused(xyz_m, albedo, spec, rough):
            surf2c = self._calc_vdir(rayo, xyz_m)
            surf2l = self._calc_ldir(xyz_m)
            normal_pred = self._normal_correct(normal_m, surf2c)
            brdf, brdf_spec, brdf_diff = self._eval_brdf_at(surf2l, surf2c, normal_pred, albedo, spec, rough)
            rgb_pred, _, _ = self._render(brdf, surf2l, normal_pred, lvis_m)
            if mode != 'train':
                rgb_diff, _, _ = self._render(brdf_diff, surf2l, normal_pred, lvis_m)
                rgb_spec, _, _ = self._render(brdf_spec, surf2l, normal_pred, lvis_m)
        else:
            sh = self._shade(xyz_m, normal_m, rayo, lvis_m, [(albedo, spec, rough)], split=(mode != 'train'))
            rgb_pred, normal_pred, rgb_diff, rgb_spec = sh['rgb'][0], sh['normal'], sh['rgb_diff'], sh['rgb_spec']
        loss_kwargs = {'mode': mode, 'pretrain': pretrain, 'gtc': rgb_m, 'rgb': rgb_pred, 'env': self._light,
                       'spec': spec, 'rough': rough}
        rgb_out = imgutil.linear2srgb(rgb_pred) if self.data_type == 'nerf' else rgb_pred
        pred = {'rgb': scatter_rows(mask, rgb_out, n), 'normal': scatter_rows(mask, normal_pred, n),
                'albedo': scatter_rows(mask, albedo, n), 'basecolor': scatter_rows(mask, basecolor, n),
                'alpha': pred_alpha, 'spec': scatter_rows(mask, spec, n), 'rough': scatter_rows(mask, rough, n),
                'ks': scatter_rows(mask, ks, n), 'xyz': scatter_rows(mask, xyz_m, n)}
        if mode != 'train':
            pred['rgb_spec'] = scatter_rows(mask, rgb_spec, n)
            pred['rgb_diff'] = scatter_rows(mask, rgb_diff, n)
        to_vis = {'id': id_, 'hw': hw}
        for k, v in pred.items():
            to_vis['pred_' + k] = v
        for k, v in gt.items():
            to_vis['gt_' + k] = v
        return pred, gt, loss_kwargs, to_vis

    def compute_loss(self, pred, gt, **kwargs):
        mode = kwargs.pop('mode')
        rgb_gt, rgb_pred = kwargs.pop('gtc'), kwargs.pop('rgb')
        linear_gt = imgutil.srgb2linear(rgb_gt) if self.data_type == 'nerf' else rgb_gt
        loss = ((linear_gt - rgb_pred) ** 2).mean(-1)
        if mode != 'train':
            return loss                                     # nfr_unit.py:415 returns the bare tensor in vali mode
        return self._numerics(loss, 'Loss'), {'rgb': loss, 'loss': loss}
