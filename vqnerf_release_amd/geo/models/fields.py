"""NeuS networks on MI355X: host-side mirror of geo/NeuS-ours2/models/fields.py (same class names,
constructor keywords, methods and state_dict keys `lin{l}.weight_g|weight_v|bias`, so reference
checkpoints `{sdf_network_fine, color_network_fine, variance_network_fine, nerf}` load unchanged).

Inference (`torch.no_grad()` / frozen parameters) runs the fused HIP kernels of csrc/neus_mlp.hip
through weight packs that are re-gathered only when a parameter changes.  When autograd needs the
graph (training), the same modules evaluate with torch ops on the GPU; see DESIGN.md "training path".
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import vqnerf_release_amd
from vqnerf_release_amd import _C
from vqnerf_release_amd.geo import packing
from vqnerf_release_amd.geo.models.embedder import get_embedder


class _Lin(nn.Module):
    """nn.Linear, optionally re-parameterised like nn.utils.weight_norm(dim=0): w = g * v / ||v||_row."""

    def __init__(self, d_in, d_out, weight_norm):
        super().__init__()
        self.weight_norm = weight_norm
        w = torch.empty(d_out, d_in)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(d_in)
        self.bias = nn.Parameter(torch.empty(d_out).uniform_(-bound, bound))
        if weight_norm:
            self.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True))
            self.weight_v = nn.Parameter(w)
        else:
            self.weight = nn.Parameter(w)

    def set_weight(self, w):
        with torch.no_grad():
            if self.weight_norm:
                self.weight_v.copy_(w)
                self.weight_g.copy_(w.norm(dim=1, keepdim=True))
            else:
                self.weight.copy_(w)

    def effective_weight(self):
        if self.weight_norm:
            return self.weight_g * self.weight_v / self.weight_v.norm(dim=1, keepdim=True)
        return self.weight

    def forward(self, x):
        return F.linear(x, self.effective_weight(), self.bias)


class _WeightNormAll(torch.autograd.Function):
    """effective weights of a list of weight-normed layers in ONE kernel launch, gradients in another (csrc/weight_norm.hip);
    the reference's per-layer norm / div / mul and their autograd are ~150 launches per step for the 13 NeuS layers."""

    @staticmethod
    def forward(ctx, n, *gv):
        import ctypes
        import numpy as np
        from vqnerf_release_amd import _C
        gs, vs = gv[:n], gv[n:]
        ws = [torch.empty_like(v) for v in vs]
        rows = np.array([v.shape[0] for v in vs], np.int32)
        cols = np.array([v.shape[1] for v in vs], np.int32)
        P = ctypes.c_void_p * n
        rc = _C.lib().vqn_weight_norm_fwd(ctypes.c_int(n), P(*[v.data_ptr() for v in vs]), P(*[g.data_ptr() for g in gs]),
                                          P(*[w.data_ptr() for w in ws]), rows.ctypes.data_as(ctypes.c_void_p),
                                          cols.ctypes.data_as(ctypes.c_void_p), _C._stream())
        _C._check(rc, 'vqn_weight_norm_fwd')
        ctx.n = n
        ctx.save_for_backward(*gv)
        return tuple(ws)

    @staticmethod
    def backward(ctx, *dws):
        import ctypes
        import numpy as np
        from vqnerf_release_amd import _C
        n = ctx.n
        gs, vs = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        dws = [torch.zeros_like(v) if d is None else d.contiguous() for d, v in zip(dws, vs)]
        dvs = [torch.empty_like(v) for v in vs]
        dgs = [torch.empty_like(g) for g in gs]
        rows = np.array([v.shape[0] for v in vs], np.int32)
        cols = np.array([v.shape[1] for v in vs], np.int32)
        P = ctypes.c_void_p * n
        rc = _C.lib().vqn_weight_norm_bwd(ctypes.c_int(n), P(*[v.data_ptr() for v in vs]), P(*[g.data_ptr() for g in gs]),
                                          P(*[d.data_ptr() for d in dws]), P(*[d.data_ptr() for d in dvs]),
                                          P(*[d.data_ptr() for d in dgs]), rows.ctypes.data_as(ctypes.c_void_p),
                                          cols.ctypes.data_as(ctypes.c_void_p), _C._stream())
        _C._check(rc, 'vqn_weight_norm_bwd')
        return (None,) + tuple(dgs) + tuple(dvs)


def effective_weights(lins):
    """[lin.effective_weight() for lin in lins]; weight-normed layers on a GPU go through the fused kernels (up to 24 per call)."""
    idx = [i for i, m in enumerate(lins) if m.weight_norm and m.weight_v.is_cuda and m.weight_v.dtype == torch.float32]
    out = [None if i in idx else m.effective_weight() for i, m in enumerate(lins)]
    for c0 in range(0, len(idx), 24):
        chunk = idx[c0:c0 + 24]
        ws = _WeightNormAll.apply(len(chunk), *[lins[i].weight_g for i in chunk], *[lins[i].weight_v for i in chunk])
        for i, w in zip(chunk, ws):
            out[i] = w
    return out


def _needs_graph(module, *tensors):
    if not torch.is_grad_enabled():
        return False
    return any(p.requires_grad for p in module.parameters()) or any(t.requires_grad for t in tensors if t is not None)


class _PackCache:
    def __init__(self):
        self.key = None
        self.value = None

    def get(self, params, device, build):
        # (the process-wide weights epoch: writes `_version` does not see -- fused Adam, graph replays; vqnerf_release_amd/__init__.py)
        key = (str(device), vqnerf_release_amd.weights_epoch()) + tuple((id(p), p._version) for p in params)
        if key != self.key:
            with torch.no_grad():
                self.value = build()
            self.key = key
        return self.value


class SDFNetwork(nn.Module):
    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, bias=0.5, scale=1,
                 geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        dims = [d_in] + [d_hidden] * n_layers + [d_out]
        self.embed_fn_fine = None
        self.multires = multires
        if multires > 0:
            self.embed_fn_fine, dims[0] = get_embedder(multires, input_dims=d_in)
        self.dims = dims
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        self.d_in = d_in
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = _Lin(dims[l], out_dim, weight_norm)
            if geometric_init:
                self._geometric_init(lin, l, dims, out_dim, bias, inside_outside)
            setattr(self, 'lin' + str(l), lin)
        self.activation = nn.Softplus(beta=100)
        self._plan = None
        self._cache = _PackCache()
        self._alt = {}                    # other matrix modes: mode -> (plan, pack cache)

    def _geometric_init(self, lin, l, dims, out_dim, bias, inside_outside):
        # sphere initialisation of IDR/NeuS (fields.py:45-63)
        w = torch.empty(out_dim, dims[l])
        b = torch.zeros(out_dim)
        last = self.num_layers - 2
        if l == last:
            sgn = -1.0 if inside_outside else 1.0
            w.normal_(sgn * np.sqrt(np.pi) / np.sqrt(dims[l]), 0.0001)
            b.fill_(-sgn * bias)
        else:
            w.normal_(0.0, np.sqrt(2) / np.sqrt(out_dim))
            if self.multires > 0 and l == 0:
                w[:, 3:] = 0.0
            elif self.multires > 0 and l in self.skip_in:
                w[:, -(dims[0] - 3):] = 0.0
        lin.set_weight(w)
        with torch.no_grad():
            lin.bias.copy_(b)

    # ---- HIP path -------------------------------------------------------------------------
    def _hip_supported(self):
        return self.d_in == 3 and self.multires > 0 and len([s for s in self.skip_in if 0 < s < self.num_layers - 1]) <= 1

    def plan(self, max_tiles=None, mode='f32'):
        """Pack plan; mode 'f16s' = packs of the split-precision kernels (csrc/neus_mlp_f16s.hip), cached beside the f32 ones."""
        if mode == 'f32':
            if self._plan is None or (max_tiles and self._plan.max_tiles < max_tiles):
                self._plan = packing.SdfPackPlan(self.dims, self.skip_in, self.multires, self.scale, max_tiles=max_tiles)
                self._cache = _PackCache()
            return self._plan
        cur = self._alt.get(mode)
        if cur is None or (max_tiles and cur[0].max_tiles < max_tiles):
            cur = (packing.SdfPackPlan(self.dims, self.skip_in, self.multires, self.scale, max_tiles=max_tiles, mode=mode), _PackCache())
            self._alt[mode] = cur
        return cur[0]

    def packs(self, max_tiles=None, mode='f32'):
        plan = self.plan(max_tiles, mode)
        cache = self._cache if mode == 'f32' else self._alt[mode][1]
        lins = [getattr(self, 'lin' + str(l)) for l in range(self.num_layers - 1)]
        params = list(self.parameters())
        return cache.get(params, params[0].device,
                         lambda: plan.pack([m.effective_weight() for m in lins], [m.bias for m in lins]))

    # ---- reference API --------------------------------------------------------------------
    def forward(self, inputs):
        inputs = inputs * self.scale
        if self.embed_fn_fine is not None:
            inputs = self.embed_fn_fine(inputs)
        x = inputs
        for l in range(self.num_layers - 1):
            if l in self.skip_in:
                x = torch.cat([x, inputs], 1) / np.sqrt(2)
            x = getattr(self, 'lin' + str(l))(x)
            if l < self.num_layers - 2:
                x = self.activation(x)
        return torch.cat([x[:, :1] / self.scale, x[:, 1:]], dim=-1)

    def _require_hip(self, x, what):
        _C.require_device(x, what)
        if not self._hip_supported():
            raise _C.VqnError(f'{what}: the fused kernels need d_in == 3, multires > 0 and at most one skip layer')

    def sdf(self, x):
        if _needs_graph(self, x):
            return self.forward(x)[:, :1]
        self._require_hip(x, 'SDFNetwork.sdf')
        wbuf, desc = self.packs()
        return _C.neus_sdf_points(desc, wbuf, pts=x.detach().float().contiguous()).reshape(-1, 1)

    def sdf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        if _needs_graph(self, x):
            x.requires_grad_(True)
            with torch.enable_grad():
                y = self.forward(x)[:, :1]
                g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True, retain_graph=True,
                                        only_inputs=True)[0]
            return g.unsqueeze(1)
        self._require_hip(x, 'SDFNetwork.gradient')
        wbuf, desc = self.packs()
        xx = x.detach().float().contiguous()
        no_col = np.zeros(packing.COL_DESC_INTS, np.int32)
        _, g, _ = _C.neus_fine_points(desc, wbuf, no_col, wbuf, pts=xx, dirs=xx)
        return g.unsqueeze(1)


class RenderingNetwork(nn.Module):
    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=0,
                 squeeze_out=True):
        super().__init__()
        self.mode = mode
        self.squeeze_out = squeeze_out
        self.d_feature = d_feature
        self.multires_view = multires_view
        dims = [d_in + d_feature] + [d_hidden] * n_layers + [d_out]
        self.embedview_fn = None
        if multires_view > 0:
            self.embedview_fn, input_ch = get_embedder(multires_view)
            dims[0] += input_ch - 3
        self.dims = dims
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            setattr(self, 'lin' + str(l), _Lin(dims[l], dims[l + 1], weight_norm))
        self.relu = nn.ReLU()
        self._plan = None
        self._cache = _PackCache()
        self._alt = {}                    # other matrix modes: mode -> (plan, pack cache)

    def max_tiles(self):
        return max((d + 31) // 32 for d in self.dims[1:-1])

    def _hip_supported(self):
        want = 3 + (3 + 6 * self.multires_view if self.mode in ('idr', 'no_normal') else 0) \
            + (3 if self.mode in ('idr', 'no_view_dir') else 0) + self.d_feature
        return self.dims[-1] == 3 and self.multires_view > 0 and self.dims[0] == want

    def packs(self, feat_tiles, mode='f32'):
        if mode == 'f32':
            if self._plan is None or self._plan.feat_tiles != feat_tiles:
                self._plan = packing.ColPackPlan(self.d_feature, self.mode, self.dims[1], self.num_layers - 2, self.dims[-1],
                                                 self.multires_view, self.squeeze_out, feat_tiles)
                self._cache = _PackCache()
            plan, cache = self._plan, self._cache
        else:
            cur = self._alt.get(mode)
            if cur is None or cur[0].feat_tiles != feat_tiles:
                cur = (packing.ColPackPlan(self.d_feature, self.mode, self.dims[1], self.num_layers - 2, self.dims[-1],
                                           self.multires_view, self.squeeze_out, feat_tiles, matrix_mode=mode), _PackCache())
                self._alt[mode] = cur
            plan, cache = cur
        lins = [getattr(self, 'lin' + str(l)) for l in range(self.num_layers - 1)]
        params = list(self.parameters())
        return cache.get(params, params[0].device,
                         lambda: plan.pack([m.effective_weight() for m in lins], [m.bias for m in lins]))

    def forward(self, points, normals, view_dirs, feature_vectors):
        if self.embedview_fn is not None:
            view_dirs = self.embedview_fn(view_dirs)
        if self.mode == 'idr':
            x = torch.cat([points, view_dirs, normals, feature_vectors], dim=-1)
        elif self.mode == 'no_view_dir':
            x = torch.cat([points, normals, feature_vectors], dim=-1)
        elif self.mode == 'no_normal':
            x = torch.cat([points, view_dirs, feature_vectors], dim=-1)
        else:
            raise ValueError(self.mode)
        for l in range(self.num_layers - 1):
            x = getattr(self, 'lin' + str(l))(x)
            if l < self.num_layers - 2:
                x = self.relu(x)
        return torch.sigmoid(x) if self.squeeze_out else x


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.register_parameter('variance', nn.Parameter(torch.tensor(float(init_val))))

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)


class NeRF(nn.Module):
    """NeRF++ background net (fields.py:176-254).  Instantiated by every shipped conf but never
    evaluated (n_outside = 0 everywhere); kept so that checkpoints' `nerf` state_dict loads."""

    def __init__(self, D=8, W=256, d_in=3, d_in_view=3, multires=0, multires_view=0, output_ch=4, skips=[4],
                 use_viewdirs=False):
        super().__init__()
        self.D, self.W, self.skips, self.use_viewdirs = D, W, skips, use_viewdirs
        self.input_ch, self.input_ch_view = 3, 3
        self.embed_fn = self.embed_fn_view = None
        if multires > 0:
            self.embed_fn, self.input_ch = get_embedder(multires, input_dims=d_in)
        if multires_view > 0:
            self.embed_fn_view, self.input_ch_view = get_embedder(multires_view, input_dims=d_in_view)
        self.pts_linears = nn.ModuleList(
            [nn.Linear(self.input_ch, W)] +
            [nn.Linear(W + self.input_ch if i in skips else W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(self.input_ch_view + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = nn.Linear(W, W)
            self.alpha_linear = nn.Linear(W, 1)
            self.rgb_linear = nn.Linear(W // 2, 3)
        else:
            self.output_linear = nn.Linear(W, output_ch)

    def forward(self, input_pts, input_views):
        if self.embed_fn is not None:
            input_pts = self.embed_fn(input_pts)
        if self.embed_fn_view is not None:
            input_views = self.embed_fn_view(input_views)
        h = input_pts
        for i, lin in enumerate(self.pts_linears):
            h = F.relu(lin(h))
            if i in self.skips:
                h = torch.cat([input_pts, h], -1)
        if not self.use_viewdirs:
            raise AssertionError('NeRF without view directions is not supported (as in the reference)')
        alpha = self.alpha_linear(h)
        h = torch.cat([self.feature_linear(h), input_views], -1)
        for lin in self.views_linears:
            h = F.relu(lin(h))
        return alpha, self.rgb_linear(h)
