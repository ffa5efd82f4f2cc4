"""ctypes binding of libvqnerf_hip.so (include/vqnerf_hip.h).

There is NO fallback: if the library is missing or a call fails this raises.  Tensors are passed as
raw device pointers (tensor.data_ptr()) plus sizes; work is enqueued on torch's current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('VQN_LIB', os.path.join(_HERE, 'lib', 'libvqnerf_hip.so'))     # VQN_LIB: diagnostic builds only
_lib = None


class VqnError(RuntimeError):
    pass


def build(verbose=False):
    """Compile csrc/*.hip for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    import subprocess
    cmd = ['make', '-C', os.path.join(_HERE, 'csrc'), '-j8']
    if not verbose:
        cmd.append('-s')
    subprocess.check_call(cmd)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VqnError(f'{LIB_PATH} is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                           '(or `make -C vqnerf_release_amd/csrc`). There is no non-HIP fallback.')
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.vqn_last_error.restype = ctypes.c_char_p
        _lib.vqn_version.restype = ctypes.c_int
    return _lib


def require_device(t, what):
    """Inference (no autograd graph) runs only on the HIP kernels: CPU tensors are an error, not a fallback."""
    if not t.is_cuda:
        raise VqnError(f'{what}: got a {t.device} tensor; the no-graph path runs on MI355X HIP kernels only '
                       '(there is no CPU fallback)')
    lib()


def _check(rc, name):
    if rc != 0:
        raise VqnError(f'{name} failed (rc={rc}): {lib().vqn_last_error().decode()}')


def _ptr(t):
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelClock:
    """Optional per-kernel device timing: HIP events recorded on the stream the kernel is launched on
    (torch's current stream).  Off by default; bench.py switches it on for the timed region.
    Events are POOLED: a process that keeps creating event pairs (two per clocked call, all alive until the summary) runs into a
    one-off stall of tens of ms when the runtime extends its event pool -- scripts/debug/launch_stall.py: 37 ms at the 2,400th live
    event, none for plain launches -- which round 3's bench had to prime its training legs past.  reset() hands the window's events
    back to the free list instead of dropping them."""
    enabled = False
    pairs = {}
    _free = []

    @classmethod
    def _event(cls):
        return cls._free.pop() if cls._free else torch.cuda.Event(enable_timing=True)

    @classmethod
    def reserve(cls, n):
        """create n events now (outside any timed region)"""
        while len(cls._free) < n:
            cls._free.append(torch.cuda.Event(enable_timing=True))

    @classmethod
    def reset(cls, enabled=True):
        for v in cls.pairs.values():
            for a, b in v:
                cls._free.append(a)
                cls._free.append(b)
        cls.enabled = enabled
        cls.pairs = {}

    @classmethod
    def summary(cls):
        """{name: (launches, total_ms)} -- call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in cls.pairs.items()}


class _clock:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        # (not while a HIP graph records: an event recorded into a capture has no time stamp; the launch is still counted)
        self.on = KernelClock.enabled and not torch.cuda.is_current_stream_capturing()
        if KernelClock.enabled and not self.on:
            KernelClock.pairs.setdefault(self.name + ' [captured]', [])
        if self.on:
            self.e0 = KernelClock._event()
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            e1 = KernelClock._event()
            e1.record()
            KernelClock.pairs.setdefault(self.name, []).append((self.e0, e1))
        return False


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise VqnError(f'{name}: expected a contiguous float32 device tensor, got {t.dtype} '
                       f'contiguous={t.is_contiguous()} device={t.device}')
    return t


# --------------------------------------------------------------------------------------
def vq_assign(x, codebook, sel_mask=None, want_quant=True, want_dist=False):
    """x [N,D], codebook [D,K] -> (idx int64 [N], quant [N,D] | None, dist [N,K] | None)."""
    _f32c(x, 'x'); _f32c(codebook, 'codebook')
    N, D = x.shape
    K = codebook.shape[1]
    assert codebook.shape[0] == D
    idx = torch.empty((N,), dtype=torch.int64, device=x.device)
    quant = torch.empty((N, D), dtype=torch.float32, device=x.device) if want_quant else None
    dist = torch.empty((N, K), dtype=torch.float32, device=x.device) if want_dist else None
    ws = None
    if sel_mask is not None:
        sel_mask = _f32c(sel_mask.reshape(-1).to(torch.float32).contiguous(), 'sel_mask')
        assert sel_mask.numel() == K
        ws = torch.empty((4,), dtype=torch.float32, device=x.device)
    with _clock('vqn_vq_assign'):
        rc = lib().vqn_vq_assign(_ptr(x), ctypes.c_int64(N), ctypes.c_int(D), _ptr(codebook), ctypes.c_int(K),
                                 _ptr(sel_mask), _ptr(ws), _ptr(idx), _ptr(quant), _ptr(dist), _stream())
    _check(rc, 'vqn_vq_assign')
    return idx, quant, dist


def vq_assign_variant(D, K, has_sel_mask=False, has_dist=False):
    """0: the f32 kernel, 1: the prefiltered kernel (vqn_vq_assign_variant)."""
    return int(lib().vqn_vq_assign_variant(ctypes.c_int(D), ctypes.c_int(K), ctypes.c_int(bool(has_sel_mask)), ctypes.c_int(bool(has_dist))))


def vq_ema_stats(x, idx, K):
    """x [N,D], idx [N] int64 -> (counts [K], dw [D,K])."""
    _f32c(x, 'x')
    N, D = x.shape
    assert idx.dtype == torch.int64 and idx.is_contiguous() and idx.numel() == N
    counts = torch.empty((K,), dtype=torch.float32, device=x.device)
    dw = torch.empty((D, K), dtype=torch.float32, device=x.device)
    L = lib()
    L.vqn_vq_ema_stats_ws_bytes.restype = ctypes.c_int64
    need = int(L.vqn_vq_ema_stats_ws_bytes(ctypes.c_int64(N), ctypes.c_int(D), ctypes.c_int(K)))
    ws = torch.empty((need // 4,), dtype=torch.float32, device=x.device) if need > 0 else None
    with _clock('vqn_vq_ema_stats'):
        rc = L.vqn_vq_ema_stats(_ptr(x), _ptr(idx), ctypes.c_int64(N), ctypes.c_int(D), ctypes.c_int(K),
                                _ptr(counts), _ptr(dw), _ptr(ws), ctypes.c_int64(need), _stream())
    _check(rc, 'vqn_vq_ema_stats')
    return counts, dw



def vq_counts(idx, K):
    """idx [N] int64 -> counts [K] float32 (= one_hot(idx, K).sum(0) without the [N,K] pass)."""
    assert idx.dtype == torch.int64 and idx.is_contiguous()
    counts = torch.empty((K,), dtype=torch.float32, device=idx.device)
    with _clock('vqn_vq_ema_stats'):
        rc = lib().vqn_vq_ema_stats(None, _ptr(idx), ctypes.c_int64(idx.numel()), ctypes.c_int(4), ctypes.c_int(K), _ptr(counts),
                                    None, None, ctypes.c_int64(0), _stream())
    _check(rc, 'vqn_vq_ema_stats')
    return counts


def vq_ste_loss(x, quant, want_ste=True):
    """x, quant [N,D] -> (x + (quant - x) [N,D] or None, mean((quant - x)^2) scalar tensor), one fused pass."""
    _f32c(x, 'x'); _f32c(quant, 'quant')
    assert x.shape == quant.shape
    n = x.numel()
    ste = torch.empty_like(x) if want_ste else None
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    ws = torch.empty((1024,), dtype=torch.float32, device=x.device)
    with _clock('vqn_vq_ste_loss'):
        rc = lib().vqn_vq_ste_loss(_ptr(x), _ptr(quant), ctypes.c_int64(n), ctypes.c_float(1.0 / n if n else 0.0), _ptr(ste),
                                   _ptr(loss), _ptr(ws), _stream())
    _check(rc, 'vqn_vq_ste_loss')
    return ste, loss

def vq_ema_update(counts, dw, codebook, decay, eps, ema_cs, ema_dw):
    """One launch for the EMA codebook move (both averages' state updated in place) -> update [D,K]."""
    _f32c(counts, 'counts'); _f32c(dw, 'dw'); _f32c(codebook, 'codebook')
    D, K = codebook.shape
    for m in (ema_cs, ema_dw):
        _f32c(m.hidden, 'hidden'); _f32c(m.average, 'average')
        assert m.counter.dtype == torch.int64 and m.counter.is_cuda
    update = torch.empty_like(codebook)
    with _clock('vqn_vq_ema_update'):
        rc = lib().vqn_vq_ema_update(_ptr(counts), _ptr(dw), _ptr(codebook), ctypes.c_int(D), ctypes.c_int(K), ctypes.c_double(decay),
                                     ctypes.c_float(eps), _ptr(ema_cs.hidden), _ptr(ema_cs.average), _ptr(ema_cs.counter),
                                     _ptr(ema_dw.hidden), _ptr(ema_dw.average), _ptr(ema_dw.counter), _ptr(update), _stream())
    _check(rc, 'vqn_vq_ema_update')
    return update


def _loss_args(rgb_pred, vq_rgb, rgb_gt, z, spec, rough, nerf, w):
    for t, n in ((rgb_pred, 'rgb_pred'), (vq_rgb, 'vq_rgb'), (rgb_gt, 'rgb_gt')):
        _f32c(t, n)
    for t, n in ((z, 'z'), (spec, 'spec'), (rough, 'rough')):
        if t is not None:
            _f32c(t, n)
    N = rgb_pred.shape[0]
    D = 0 if z is None else z.shape[1]
    return (_ptr(rgb_pred), _ptr(vq_rgb), _ptr(rgb_gt), _ptr(z), _ptr(spec), _ptr(rough), ctypes.c_int64(N), ctypes.c_int(D),
            ctypes.c_int(1 if nerf else 0)) + tuple(ctypes.c_float(float(w[k])) for k in ('rgb', 'chr', 'smooth', 'alpha', 'thres', 'lambert'))


def decomp_loss_fwd(rgb_pred, vq_rgb, rgb_gt, z, spec, rough, nerf, w):
    """-> terms [N,5] (rgb, vqrgb, chromaticity, chr_smooth, lambert); w: dict of the six scalars (see include/vqnerf_hip.h)."""
    terms = torch.empty((rgb_pred.shape[0], 5), dtype=torch.float32, device=rgb_pred.device)
    with _clock('vqn_decomp_loss_fwd'):
        rc = lib().vqn_decomp_loss_fwd(*_loss_args(rgb_pred, vq_rgb, rgb_gt, z, spec, rough, nerf, w), _ptr(terms), _stream())
    _check(rc, 'vqn_decomp_loss_fwd')
    return terms


def decomp_loss_bwd(rgb_pred, vq_rgb, rgb_gt, z, spec, rough, nerf, w, g_terms):
    _f32c(g_terms, 'g_terms')
    g_pred, g_vq = torch.empty_like(rgb_pred), torch.empty_like(vq_rgb)
    g_z = None if z is None else torch.empty_like(z)
    g_spec = None if spec is None else torch.empty_like(spec)
    with _clock('vqn_decomp_loss_bwd'):
        rc = lib().vqn_decomp_loss_bwd(*_loss_args(rgb_pred, vq_rgb, rgb_gt, z, spec, rough, nerf, w), _ptr(g_terms), _ptr(g_pred), _ptr(g_vq),
                                       _ptr(g_z), _ptr(g_spec), _stream())
    _check(rc, 'vqn_decomp_loss_bwd')
    return g_pred, g_vq, g_z, g_spec


def codebook_prep(raw, g=None, eps=1e-6):
    """clip-with-identity-gradient to [0, 1] + l2-normalise every column of raw [D, K] (g None), or the gradient wrt raw for g [D, K]."""
    _f32c(raw, 'raw')
    if g is not None:
        _f32c(g, 'g')
    out = torch.empty_like(raw)
    with _clock('vqn_codebook_prep'):
        rc = lib().vqn_codebook_prep(_ptr(raw), _ptr(g), ctypes.c_int(raw.shape[0]), ctypes.c_int(raw.shape[1]), ctypes.c_float(eps), _ptr(out), _stream())
    _check(rc, 'vqn_codebook_prep')
    return out


def sim_smooth_fwd(cb, weight):
    _f32c(cb, 'codebook')
    out = torch.empty(4, dtype=torch.float32, device=cb.device)
    with _clock('vqn_sim_smooth_fwd'):
        rc = lib().vqn_sim_smooth_fwd(_ptr(cb), ctypes.c_int(cb.shape[0]), ctypes.c_int(cb.shape[1]), ctypes.c_float(weight), _ptr(out), _stream())
    _check(rc, 'vqn_sim_smooth_fwd')
    return out


def sim_smooth_bwd(cb, fwd4, g_loss, weight):
    _f32c(cb, 'codebook'); _f32c(fwd4, 'fwd4'); _f32c(g_loss, 'g_loss')
    g = torch.empty_like(cb)
    with _clock('vqn_sim_smooth_bwd'):
        rc = lib().vqn_sim_smooth_bwd(_ptr(cb), _ptr(fwd4), _ptr(g_loss), ctypes.c_int(cb.shape[0]), ctypes.c_int(cb.shape[1]), ctypes.c_float(weight),
                                      _ptr(g), _stream())
    _check(rc, 'vqn_sim_smooth_bwd')
    return g


def l2_normalize_rows_bwd(x, g, eps=1e-6):
    """Gradient of l2_normalize_rows at x [N, D] for the incoming g [N, D]: one pass (vqn_l2_normalize_rows_bwd)."""
    _f32c(x, 'x'); _f32c(g, 'g')
    if x.shape != g.shape or x.dim() != 2:
        raise VqnError('l2_normalize_rows_bwd: x and g must be [N, D]')
    gx = torch.empty_like(x)
    with _clock('vqn_l2_normalize_rows_bwd'):
        rc = lib().vqn_l2_normalize_rows_bwd(_ptr(x), _ptr(g), ctypes.c_int64(x.shape[0]), ctypes.c_int(x.shape[1]), ctypes.c_float(eps), _ptr(gx),
                                             _stream())
    _check(rc, 'vqn_l2_normalize_rows_bwd')
    return gx


def vq_ste_loss_bwd(x, quant, g_ste, g_loss):
    """g_ste + (x - quant) * (g_loss * 2 / numel) in one pass (g_ste may be None; g_loss a 0-dim device tensor)."""
    _f32c(x, 'x'); _f32c(quant, 'quant'); _f32c(g_loss, 'g_loss')
    if g_ste is not None:
        _f32c(g_ste, 'g_ste')
    gx = torch.empty_like(x)
    with _clock('vqn_vq_ste_loss_bwd'):
        rc = lib().vqn_vq_ste_loss_bwd(_ptr(x), _ptr(quant), _ptr(g_ste), _ptr(g_loss), ctypes.c_int64(x.numel()), _ptr(gx), _stream())
    _check(rc, 'vqn_vq_ste_loss_bwd')
    return gx


def vq_train_bwd(z, xnorm, quant, g_ste, g_loss, eps=1e-6, loss_post=1.0):
    """Backward of the training quantiser in one pass (vqn_vq_train_bwd): straight-through + commitment adjoint, then the l2-normalise
    backward of z."""
    for t in (z, xnorm, quant, g_loss):
        _f32c(t, 'tensor')
    if g_ste is not None:
        _f32c(g_ste, 'g_ste')
    gz = torch.empty_like(z)
    with _clock('vqn_vq_train_bwd'):
        rc = lib().vqn_vq_train_bwd(_ptr(z), _ptr(xnorm), _ptr(quant), _ptr(g_ste), _ptr(g_loss), ctypes.c_float(loss_post), ctypes.c_int64(z.shape[0]),
                                    ctypes.c_int(z.shape[1]), ctypes.c_float(eps), _ptr(gz), _stream())
    _check(rc, 'vqn_vq_train_bwd')
    return gz


def l2_normalize_rows(x, eps=1e-6):
    """x [N,D] -> x / sqrt(max(sum_d x^2, eps)) row by row, in the defined summation order of vqn_vq_assign's |x|^2."""
    _f32c(x, 'x')
    y = torch.empty_like(x)
    with _clock('vqn_l2_normalize_rows'):
        rc = lib().vqn_l2_normalize_rows(_ptr(x), ctypes.c_int64(x.shape[0]), ctypes.c_int(x.shape[1]), ctypes.c_float(eps), _ptr(y), _stream())
    _check(rc, 'vqn_l2_normalize_rows')
    return y


def vq_quantize_rows(z, codebook, sel_mask=None, eps=1e-6, want_ste=True, want_xnorm=False, loss_post=1.0):
    """Fused inference path: z [N,D] un-normalised, codebook [D,K] -> (idx int64 [N], ste [N,D] | None, mean((q - z^)^2) scalar
    tensor, counts [K]) in one pass over the rows (l2-normalise, nearest code, straight-through output, commitment term, usage)."""
    _f32c(z, 'z'); _f32c(codebook, 'codebook')
    N, D = z.shape
    K = codebook.shape[1]
    assert codebook.shape[0] == D
    dev = z.device
    idx = torch.empty((N,), dtype=torch.int64, device=dev)
    ste = torch.empty((N, D), dtype=torch.float32, device=dev) if want_ste else None
    loss = torch.empty((), dtype=torch.float32, device=dev)
    counts = torch.empty((K,), dtype=torch.float32, device=dev)
    ws = torch.empty((4096,), dtype=torch.float32, device=dev)
    if sel_mask is not None:
        sel_mask = _f32c(sel_mask.reshape(-1).to(torch.float32).contiguous(), 'sel_mask')
        assert sel_mask.numel() == K
    n = N * D
    if want_xnorm:                                         # the training form: the normalised rows are kept (EMA statistics, backward)
        xnorm = torch.empty((N, D), dtype=torch.float32, device=dev)
        with _clock('vqn_vq_quantize_rows_train'):
            rc = lib().vqn_vq_quantize_rows_train(_ptr(z), ctypes.c_int64(N), ctypes.c_int(D), _ptr(codebook), ctypes.c_int(K), _ptr(sel_mask),
                                                  ctypes.c_float(eps), ctypes.c_float(1.0 / n if n else 0.0), ctypes.c_float(loss_post), _ptr(ws), _ptr(idx),
                                                  _ptr(ste), _ptr(loss), _ptr(counts), _ptr(xnorm), _stream())
        _check(rc, 'vqn_vq_quantize_rows_train')
        return idx, ste, loss, counts, xnorm
    with _clock('vqn_vq_quantize_rows'):
        rc = lib().vqn_vq_quantize_rows(_ptr(z), ctypes.c_int64(N), ctypes.c_int(D), _ptr(codebook), ctypes.c_int(K), _ptr(sel_mask),
                                        ctypes.c_float(eps), ctypes.c_float(1.0 / n if n else 0.0), _ptr(ws), _ptr(idx), _ptr(ste),
                                        _ptr(loss), _ptr(counts), _stream())
    _check(rc, 'vqn_vq_quantize_rows')
    return idx, ste, loss, counts


# --------------------------------------------------------------------------------------
# fused NeuS networks (csrc/neus_mlp.hip)
_scratch = {}


def _i32(desc):
    import numpy as np
    d = np.ascontiguousarray(desc, dtype=np.int32)
    return d, d.ctypes.data_as(ctypes.c_void_p)


def neus_sdf_points(sdf_desc, wbuf_sdf, rays_o=None, rays_d=None, z=None, pts=None, mode='f32', pack=None):
    """SDF value at ray samples (rays_o/rays_d [B,3], z [B,S]) or at explicit pts [P,3] -> [P].  pack: a NeusPackHandle (its
    descriptor and SDF buffer are used, `mode` names the engine it was created for)."""
    if pack is not None:
        dp, wp = pack.sdf_desc, pack.sdf_wbuf
    else:
        _f32c(wbuf_sdf, 'wbuf_sdf')
        d, dp = _i32(sdf_desc)
        wp = _ptr(wbuf_sdf)
    if pts is not None:
        _f32c(pts, 'pts'); P, S = pts.shape[0], 1
        dev = pts.device
    else:
        _f32c(rays_o, 'rays_o'); _f32c(rays_d, 'rays_d'); _f32c(z, 'z')
        B, S = z.shape
        P = B * S
        dev = z.device
    out = torch.empty((P,), dtype=torch.float32, device=dev)
    entry = {'f32': 'vqn_neus_sdf_points', 'f16s': 'vqn_neus_sdf_points_f16s', 'x3': 'vqn_neus_sdf_points_x3'}[mode]
    with _clock(entry):
        rc = getattr(lib(), entry)(dp, wp, _ptr(rays_o), _ptr(rays_d), _ptr(z), _ptr(pts),
                                   ctypes.c_int64(P), ctypes.c_int(S), _ptr(out), _stream())
    _check(rc, entry)
    return out


def neus_fine_points(sdf_desc, wbuf_sdf, col_desc, wbuf_col, rays_o=None, rays_d=None, z=None, pts=None, dirs=None, mode='f32'):
    """sdf [P], d sdf/d x [P,3], rgb [P,3] at ray samples or explicit (pts, dirs)."""
    _f32c(wbuf_sdf, 'wbuf_sdf'); _f32c(wbuf_col, 'wbuf_col')
    sd, sdp = _i32(sdf_desc)
    cd, cdp = _i32(col_desc)
    if pts is not None:
        _f32c(pts, 'pts'); _f32c(dirs, 'dirs'); P, S = pts.shape[0], 1
        dev = pts.device
    else:
        _f32c(rays_o, 'rays_o'); _f32c(rays_d, 'rays_d'); _f32c(z, 'z')
        B, S = z.shape
        P = B * S
        dev = z.device
    L = lib()
    L.vqn_neus_fine_scratch_bytes.restype = ctypes.c_int64
    need = int(L.vqn_neus_fine_scratch_bytes(sdp))
    if need <= 0:
        raise VqnError('vqn_neus_fine_scratch_bytes: invalid SDF descriptor')
    key = (str(dev), torch.cuda.current_stream().cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty((need,), dtype=torch.uint8, device=dev)
        _scratch[key] = buf
    sdf = torch.empty((P,), dtype=torch.float32, device=dev)
    grad = torch.empty((P, 3), dtype=torch.float32, device=dev)
    rgb = torch.empty((P, 3), dtype=torch.float32, device=dev)
    entry = {'f32': 'vqn_neus_fine_points', 'f16s': 'vqn_neus_fine_points_f16s', 'x3': 'vqn_neus_fine_points_x3'}[mode]
    with _clock(entry):
        rc = getattr(L, entry)(sdp, _ptr(wbuf_sdf), cdp, _ptr(wbuf_col), _ptr(rays_o), _ptr(rays_d), _ptr(z),
                               _ptr(pts), _ptr(dirs), ctypes.c_int64(P), ctypes.c_int(S), _ptr(buf),
                               ctypes.c_int64(buf.numel()), _ptr(sdf), _ptr(grad), _ptr(rgb), _stream())
    _check(rc, entry)
    return sdf, grad, rgb


class NeusPackHandle:
    """vqn_neus_pack_create / _update / _destroy: the packs of one (SDFNetwork, RenderingNetwork) shape built by the library from
    tables of device pointers to the effective weights [out, in] and biases (one gather launch per network and update)."""

    def __init__(self, sdf_dims, sdf_skip, multires, scale, col_mode, col_d_hidden, col_n_layers, multires_view, squeeze_out, engine):
        L = lib()
        for f in ('vqn_neus_pack_sdf_desc', 'vqn_neus_pack_col_desc', 'vqn_neus_pack_sdf_wbuf', 'vqn_neus_pack_col_wbuf'):
            getattr(L, f).restype = ctypes.c_void_p
        self.h = ctypes.c_void_p()
        dims = (ctypes.c_int32 * len(sdf_dims))(*sdf_dims)
        _check(L.vqn_neus_pack_create(dims, ctypes.c_int(len(sdf_dims) - 1), ctypes.c_int(sdf_skip), ctypes.c_int(multires), ctypes.c_float(scale),
                                      ctypes.c_int(col_mode), ctypes.c_int(col_d_hidden), ctypes.c_int(col_n_layers), ctypes.c_int(multires_view),
                                      ctypes.c_int(int(squeeze_out)), ctypes.c_int(engine), ctypes.byref(self.h)), 'vqn_neus_pack_create')
        self.sdf_desc, self.col_desc = ctypes.c_void_p(L.vqn_neus_pack_sdf_desc(self.h)), ctypes.c_void_p(L.vqn_neus_pack_col_desc(self.h))
        self.sdf_wbuf, self.col_wbuf = ctypes.c_void_p(L.vqn_neus_pack_sdf_wbuf(self.h)), ctypes.c_void_p(L.vqn_neus_pack_col_wbuf(self.h))

    def update(self, W, b, Wc, bc):
        for t in W + b + Wc + bc:
            _f32c(t, 'weight / bias')
        arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        with _clock('vqn_neus_pack_update'):
            rc = lib().vqn_neus_pack_update(self.h, arr(W), arr(b), arr(Wc), arr(bc), _stream())
        _check(rc, 'vqn_neus_pack_update')

    def __del__(self):
        try:
            if self.h:
                lib().vqn_neus_pack_destroy(self.h)
        except Exception:      # noqa: BLE001  (interpreter shutdown)
            pass


def neus_train_fwd(sdf_desc, wbuf_sdf, col_desc, wbuf_col, pts, dirs, saved, e_tiles, outf_tiles, extr_tiles, out_sdf, out_n, out_rgb, pack=None):
    """Training forward of the NeuS core at explicit (pts, dirs): the two-image fine kernel, which also leaves the tensors the backward
    tile programs read (`saved`: [E, OUTF, EXTR, U_1.., GH_0.., C_1..], tile format).  Fills out_sdf [P,1], out_n [P,3], out_rgb [P,3].
    pack: a NeusPackHandle of the exact-split engine -> vqn_neus_train_fwd_x3 on its packs (the descriptor / buffer arguments unused)."""
    _f32c(pts, 'pts'); _f32c(dirs, 'dirs')
    for t in saved + [out_sdf, out_n, out_rgb]:
        _f32c(t, 'saved tensor')
    if pack is not None:
        sdp, cdp, wsp, wcp, entry = pack.sdf_desc, pack.col_desc, pack.sdf_wbuf, pack.col_wbuf, 'vqn_neus_train_fwd_x3'
    else:
        _f32c(wbuf_sdf, 'wbuf_sdf'); _f32c(wbuf_col, 'wbuf_col')
        sd, sdp = _i32(sdf_desc)
        cd, cdp = _i32(col_desc)
        wsp, wcp, entry = _ptr(wbuf_sdf), _ptr(wbuf_col), 'vqn_neus_train_fwd'
    P, dev = pts.shape[0], pts.device
    L = lib()
    L.vqn_neus_fine_scratch_bytes.restype = ctypes.c_int64
    need = int(L.vqn_neus_fine_scratch_bytes(sdp))
    if need <= 0:
        raise VqnError('vqn_neus_fine_scratch_bytes: invalid SDF descriptor')
    key = (str(dev), torch.cuda.current_stream().cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty((need,), dtype=torch.uint8, device=dev)
        _scratch[key] = buf
    ptrs = (ctypes.c_void_p * len(saved))(*[t.data_ptr() for t in saved])
    with _clock(entry):
        rc = getattr(L, entry)(sdp, wsp, cdp, wcp, _ptr(pts), _ptr(dirs), ctypes.c_int64(P), _ptr(buf),
                               ctypes.c_int64(buf.numel()), ptrs, ctypes.c_int(len(saved)), ctypes.c_int(e_tiles),
                               ctypes.c_int(outf_tiles), ctypes.c_int(extr_tiles), _ptr(out_sdf), _ptr(out_n), _ptr(out_rgb), _stream())
    _check(rc, entry)


def pack_x3_gather(flat, gidx, n_steps, fidx=None):
    """bf16 piece triples [n_steps, 3, 64, 8] (as an int16 tensor) of flat[gidx] -- gather + exact split in one launch.  With `fidx`
    (a contiguous int32 index tensor on flat's device): also flat[fidx], in the same launch -> (pieces, f32 images)."""
    _f32c(flat, 'flat')
    if gidx.dtype != torch.int32 or not gidx.is_contiguous() or gidx.numel() != n_steps * 512:
        raise VqnError('pack_x3_gather: gidx must be a contiguous int32 tensor of n_steps * 512 entries')
    out = torch.empty((n_steps, 3, 64, 8), dtype=torch.int16, device=flat.device)
    if fidx is None:
        with _clock('vqn_pack_x3_gather'):
            rc = lib().vqn_pack_x3_gather(_ptr(flat), _ptr(gidx), ctypes.c_int64(n_steps), _ptr(out), _stream())
        _check(rc, 'vqn_pack_x3_gather')
        return out
    if fidx.dtype != torch.int32 or not fidx.is_contiguous() or fidx.device != flat.device:
        raise VqnError('pack_x3_gather: fidx must be a contiguous int32 tensor on the device of flat')
    wf = torch.empty(fidx.shape, dtype=torch.float32, device=flat.device)
    with _clock('vqn_pack_x3_gather'):
        rc = lib().vqn_pack_x3_gather2(_ptr(flat), _ptr(gidx), ctypes.c_int64(n_steps), _ptr(out), _ptr(fidx), ctypes.c_int64(fidx.numel()),
                                       _ptr(wf), _stream())
    _check(rc, 'vqn_pack_x3_gather2')
    return out, wf


def neus_train_bwd_x3(desc, wbuf_pieces, wbuf_f32, pts, g_rgb, rgb, g_n, g_sdf, saved, outs):
    """neus_train_bwd on the exact-split engine (csrc/neus_train_bwd_x3.hip)."""
    _f32c(wbuf_f32, 'wbuf_f32'); _f32c(pts, 'pts'); _f32c(g_rgb, 'g_rgb')
    for t in saved + outs + [t for t in (rgb, g_n, g_sdf) if t is not None]:
        _f32c(t, 'tensor')
    d, dp = _i32(desc)
    P, dev = pts.shape[0], pts.device
    L = lib()
    L.vqn_neus_train_bwd_x3_scratch_bytes.restype = ctypes.c_int64
    need = int(L.vqn_neus_train_bwd_x3_scratch_bytes(dp))
    if need <= 0:
        raise VqnError('vqn_neus_train_bwd_x3_scratch_bytes: invalid descriptor')
    key = (str(dev), torch.cuda.current_stream().cuda_stream, 'bwd')
    buf = _scratch.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty((need,), dtype=torch.uint8, device=dev)
        _scratch[key] = buf
    sp = (ctypes.c_void_p * len(saved))(*[t.data_ptr() for t in saved])
    op = (ctypes.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
    with _clock('vqn_neus_train_bwd_x3'):
        rc = L.vqn_neus_train_bwd_x3(dp, _ptr(wbuf_pieces), _ptr(wbuf_f32), _ptr(pts), _ptr(g_rgb), _ptr(rgb), _ptr(g_n), _ptr(g_sdf),
                                     ctypes.c_int64(P), _ptr(buf), ctypes.c_int64(buf.numel()), sp, ctypes.c_int(len(saved)), op,
                                     ctypes.c_int(len(outs)), _stream())
    _check(rc, 'vqn_neus_train_bwd_x3')


def neus_train_bwd(desc, wbuf, pts, g_rgb, rgb, g_n, g_sdf, saved, outs):
    """Backward of the NeuS core in one launch (csrc/neus_train_bwd.hip): fills `outs` ([DC_0.., GOUTF, ED, UD_1.., AB_0..]) from the
    incoming adjoints and the forward's `saved` tensors ([U_1.., GH_0.., C_1..]); rgb / g_n / g_sdf may be None."""
    _f32c(wbuf, 'wbuf'); _f32c(pts, 'pts'); _f32c(g_rgb, 'g_rgb')
    for t in saved + outs + [t for t in (rgb, g_n, g_sdf) if t is not None]:
        _f32c(t, 'tensor')
    d, dp = _i32(desc)
    P, dev = pts.shape[0], pts.device
    L = lib()
    L.vqn_neus_train_bwd_scratch_bytes.restype = ctypes.c_int64
    need = int(L.vqn_neus_train_bwd_scratch_bytes(dp))
    if need <= 0:
        raise VqnError('vqn_neus_train_bwd_scratch_bytes: invalid descriptor')
    key = (str(dev), torch.cuda.current_stream().cuda_stream, 'bwd')
    buf = _scratch.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty((need,), dtype=torch.uint8, device=dev)
        _scratch[key] = buf
    sp = (ctypes.c_void_p * len(saved))(*[t.data_ptr() for t in saved])
    op = (ctypes.c_void_p * len(outs))(*[t.data_ptr() for t in outs])
    with _clock('vqn_neus_train_bwd'):
        rc = L.vqn_neus_train_bwd(dp, _ptr(wbuf), _ptr(pts), _ptr(g_rgb), _ptr(rgb), _ptr(g_n), _ptr(g_sdf), ctypes.c_int64(P), _ptr(buf),
                                  ctypes.c_int64(buf.numel()), sp, ctypes.c_int(len(saved)), op, ctypes.c_int(len(outs)), _stream())
    _check(rc, 'vqn_neus_train_bwd')


def multi_copy(dsts, srcs):
    """dsts[i].copy_(srcs[i]) for contiguous f32 tensors of equal element counts, all in ONE launch (vqn_multi_copy)."""
    k = len(dsts)
    if k == 0:
        return
    import numpy as np
    for d, s in zip(dsts, srcs):
        _f32c(d, 'dst'); _f32c(s, 'src')
        if d.numel() != s.numel():
            raise VqnError('multi_copy: element counts differ')
    sp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in srcs])
    dp = (ctypes.c_void_p * k)(*[t.data_ptr() for t in dsts])
    n = np.array([t.numel() for t in dsts], np.int64)
    with _clock('vqn_multi_copy'):
        rc = lib().vqn_multi_copy(ctypes.c_int(k), sp, dp, n.ctypes.data_as(ctypes.c_void_p), _stream())
    _check(rc, 'vqn_multi_copy')


# --------------------------------------------------------------------------------------
# per-ray NeuS kernels (csrc/neus_rays.hip)
def neus_upsample(rays_o, rays_d, z, sdf, r_limit, inv_s, u):
    _f32c(rays_o, 'rays_o'); _f32c(rays_d, 'rays_d'); _f32c(z, 'z'); _f32c(sdf, 'sdf'); _f32c(u, 'u')
    B, n = z.shape
    m = u.numel()
    z_new = torch.empty((B, m), dtype=torch.float32, device=z.device)
    with _clock('vqn_neus_upsample'):
        rc = lib().vqn_neus_upsample(_ptr(rays_o), _ptr(rays_d), _ptr(z), _ptr(sdf), ctypes.c_int64(B), ctypes.c_int(n),
                                     ctypes.c_float(r_limit), ctypes.c_float(inv_s), _ptr(u), ctypes.c_int(m),
                                     _ptr(z_new), _stream())
    _check(rc, 'vqn_neus_upsample')
    return z_new


def neus_merge(z, sdf, z_new, sdf_new):
    _f32c(z, 'z'); _f32c(z_new, 'z_new')
    B, n = z.shape
    m = z_new.shape[1]
    z_out = torch.empty((B, n + m), dtype=torch.float32, device=z.device)
    sdf_out = None
    if sdf is not None and sdf_new is not None:
        _f32c(sdf, 'sdf'); _f32c(sdf_new, 'sdf_new')
        sdf_out = torch.empty((B, n + m), dtype=torch.float32, device=z.device)
    with _clock('vqn_neus_merge'):
        rc = lib().vqn_neus_merge(_ptr(z), _ptr(sdf if sdf_out is not None else None), _ptr(z_new),
                                  _ptr(sdf_new if sdf_out is not None else None), ctypes.c_int64(B), ctypes.c_int(n),
                                  ctypes.c_int(m), _ptr(z_out), _ptr(sdf_out), _stream())
    _check(rc, 'vqn_neus_merge')
    return z_out, sdf_out


def neus_section_mids(z, sample_dist, sample_dist_per_ray=None):
    _f32c(z, 'z')
    B, n = z.shape
    mid = torch.empty_like(z)
    dists = torch.empty_like(z)
    if sample_dist_per_ray is not None:
        sample_dist_per_ray = _f32c(sample_dist_per_ray.reshape(-1).contiguous(), 'sample_dist_per_ray')
    with _clock('vqn_neus_section_mids'):
        rc = lib().vqn_neus_section_mids(_ptr(z), ctypes.c_int64(B), ctypes.c_int(n), ctypes.c_float(float(sample_dist)),
                                         _ptr(sample_dist_per_ray), _ptr(mid), _ptr(dists), _stream())
    _check(rc, 'vqn_neus_section_mids')
    return mid, dists


def neus_composite_fwd(rays_o, rays_d, mid_z, dists, sdf, grad, rgb, inv_s, background_rgb, radius,
                       cos_anneal_ratio, want_alpha=False):
    for n_, t in (('rays_o', rays_o), ('rays_d', rays_d), ('mid_z', mid_z), ('dists', dists), ('sdf', sdf),
                  ('grad', grad), ('rgb', rgb), ('inv_s', inv_s)):
        _f32c(t, n_)
    B, n = mid_z.shape
    dev = mid_z.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    out = dict(color=f(B, 3), weights=f(B, n), cdf=f(B, n), inside_sphere=f(B, n), surf=f(B, 3), depth=f(B, 1),
               weight_sum=f(B, 1), weight_max=f(B, 1), gerr=f(B, 2))
    alpha = f(B, n) if want_alpha else None
    if background_rgb is not None:
        background_rgb = _f32c(background_rgb.reshape(-1)[:3].contiguous(), 'background_rgb')
    with _clock('vqn_neus_composite_fwd'):
        rc = lib().vqn_neus_composite_fwd(_ptr(rays_o), _ptr(rays_d), _ptr(mid_z), _ptr(dists), _ptr(sdf), _ptr(grad),
                                          _ptr(rgb), _ptr(inv_s), _ptr(background_rgb), ctypes.c_int64(B), ctypes.c_int(n),
                                          ctypes.c_float(float(radius)), ctypes.c_float(float(cos_anneal_ratio)),
                                          _ptr(out['color']), _ptr(out['weights']), _ptr(out['cdf']),
                                          _ptr(out['inside_sphere']), _ptr(out['surf']), _ptr(out['depth']),
                                          _ptr(out['weight_sum']), _ptr(out['weight_max']), _ptr(out['gerr']),
                                          _ptr(alpha), _stream())
    _check(rc, 'vqn_neus_composite_fwd')
    if want_alpha:
        out['alpha'] = alpha
    return out


# --------------------------------------------------------------------------------------
# reflectance path (csrc/mlp_chain.hip, csrc/brdf_shade.hip)
def mlp_chain_fwd(desc, wbuf, x, out_widths, mode='f32'):
    """Run a layer program (decomp/packing.py) over x [N, in_stride]; returns one [N, w] tensor per output slot.
    mode 'f16s': the split-precision kernel (program and pack must come from ChainBuilder(mode='f16s'))."""
    assert mode in ('f32', 'f16s')
    entry = 'vqn_mlp_chain_fwd' if mode == 'f32' else 'vqn_mlp_chain_fwd_f16s'
    _f32c(wbuf, 'wbuf'); _f32c(x, 'x')
    d, dp = _i32(desc)
    N = x.shape[0]
    assert x.shape[1] == int(d[8]), (x.shape, int(d[8]))
    outs = [torch.empty((N, w), dtype=torch.float32, device=x.device) for w in out_widths]
    args = []
    for i in range(4):
        if i < len(outs):
            args += [_ptr(outs[i]), ctypes.c_int(out_widths[i])]
        else:
            args += [ctypes.c_void_p(0), ctypes.c_int(0)]
    with _clock(entry):
        rc = getattr(lib(), entry)(dp, _ptr(wbuf), _ptr(x), ctypes.c_int64(N), *args, _stream())
    _check(rc, entry)
    return outs


def linear2srgb(x):
    """clip to [0, 1] + the sRGB transfer curve in one pass (vqn_linear2srgb)."""
    x = x.contiguous()
    _f32c(x, 'x')
    y = torch.empty_like(x)
    with _clock('vqn_linear2srgb'):
        rc = lib().vqn_linear2srgb(_ptr(x), ctypes.c_int64(x.numel()), _ptr(y), _stream())
    _check(rc, 'vqn_linear2srgb')
    return y


def vq_codebook_frags(codebook):
    """codebook [256, K <= 64] -> the B-fragment + |c|^2 image the fused reflectance kernel reads (vqn_vq_codebook_frags)."""
    _f32c(codebook, 'codebook')
    D, K = codebook.shape
    frags = torch.empty(((1 if K <= 16 else (2 if K <= 32 else 4)) * (16 * 64 * 4 + 16),), dtype=torch.float32, device=codebook.device)
    with _clock('vqn_vq_codebook_frags'):
        rc = lib().vqn_vq_codebook_frags(_ptr(codebook), ctypes.c_int(D), ctypes.c_int(K), _ptr(frags), _stream())
    _check(rc, 'vqn_vq_codebook_frags')
    return frags


def mlp_chain_vq_fwd(desc_a, wbuf_a, widths_a, desc_b, wbuf_b, widths_b, x, frags, K, eps=1e-6, want_z=False, want_ste=False):
    """Program A (encoder + heads, z through slot 0) -> VQ step on z in LDS -> program B (heads on the straight-through rows) in one
    launch (vqn_mlp_chain_vq_fwd).  Returns (outs_a, outs_b, idx, ste, loss, counts); outs_a[0] (z) is None unless want_z, ste
    (the straight-through rows [N, 256]) None unless want_ste."""
    _f32c(wbuf_a, 'wbuf_a'); _f32c(wbuf_b, 'wbuf_b'); _f32c(x, 'x'); _f32c(frags, 'frags')
    da, dpa = _i32(desc_a)
    db, dpb = _i32(desc_b)
    N = x.shape[0]
    dev = x.device
    assert x.shape[1] == int(da[8]), (x.shape, int(da[8]))
    outs_a = [torch.empty((N, w), dtype=torch.float32, device=dev) if (i > 0 or want_z) else None for i, w in enumerate(widths_a)]
    outs_b = [torch.empty((N, w), dtype=torch.float32, device=dev) for w in widths_b]
    tab = lambda ts: (ctypes.c_void_p * 4)(*[(0 if (i >= len(ts) or ts[i] is None) else ts[i].data_ptr()) for i in range(4)])
    lds = lambda ws: (ctypes.c_int32 * 4)(*[(ws[i] if i < len(ws) else 0) for i in range(4)])
    idx = torch.empty((N,), dtype=torch.int64, device=dev)
    ste = torch.empty((N, 256), dtype=torch.float32, device=dev) if want_ste else None
    loss = torch.empty((), dtype=torch.float32, device=dev)
    counts = torch.empty((K,), dtype=torch.float32, device=dev)
    ws = torch.empty((4096,), dtype=torch.float32, device=dev)
    n = N * 256
    with _clock('vqn_mlp_chain_vq_fwd'):
        rc = lib().vqn_mlp_chain_vq_fwd(dpa, _ptr(wbuf_a), dpb, _ptr(wbuf_b), _ptr(x), ctypes.c_int64(N), tab(outs_a), lds(widths_a),
                                        tab(outs_b), lds(widths_b), _ptr(frags), ctypes.c_int(K), ctypes.c_float(eps),
                                        ctypes.c_float(1.0 / n if n else 0.0), _ptr(idx), _ptr(ste), _ptr(loss), _ptr(counts), _ptr(ws), _stream())
    _check(rc, 'vqn_mlp_chain_vq_fwd')
    return outs_a, outs_b, idx, ste, loss, counts


def brdf_shade_fwd(xyz, normal, rayo, lvis, lxyz, lareas, light, materials, gamma=None, want_normal=True,
                   want_split=False, raw=False, probes=None, lvis_rows=None):
    """materials: [(albedo [N,3], spec [N,3], rough [N,1])] (1 or 2 sets).
    lvis_rows [N] int64 (optional): `lvis` is the FULL-view buffer [n_view, L] and point n reads its row lvis_rows[n].
    -> dict(rgb=[...per set], normal=..., rgb_diff=..., rgb_spec=...)."""
    for n_, t in (('xyz', xyz), ('normal', normal), ('rayo', rayo), ('lxyz', lxyz), ('lareas', lareas), ('light', light)):
        _f32c(t, n_)
    L = lareas.numel()
    assert lxyz.numel() == 3 * L and light.numel() == 3 * L
    N = xyz.shape[0]
    assert normal.shape[0] == N and rayo.shape[0] == N
    rows = None
    if lvis is not None:
        _f32c(lvis, 'lvis')
        if lvis_rows is not None:
            rows = lvis_rows
            assert rows.dtype == torch.int64 and rows.is_contiguous() and rows.is_cuda and rows.numel() == N and lvis.shape[1] == L
        else:
            assert tuple(lvis.shape) == (N, L)
    mats = []
    for (a, s, r) in materials:
        _f32c(a, 'albedo'); _f32c(s, 'spec'); _f32c(r, 'rough')
        assert tuple(a.shape) == (N, 3) and tuple(s.shape) == (N, 3) and r.numel() == N
        mats += [a, s, r]
    while len(mats) < 6:
        mats.append(None)
    dev = xyz.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    rgb = [f(N, 3) for _ in materials]
    nout = f(N, 3) if want_normal else None
    rd, rs = (f(N, 3), f(N, 3)) if want_split else (None, None)
    if gamma is not None:
        gamma = _f32c(gamma.reshape(-1).contiguous(), 'gamma')
    n_probes, rgb_probes = 0, None
    if probes is not None:
        probes = _f32c(probes.reshape(-1, L, 3).contiguous(), 'probes')
        n_probes = probes.shape[0]
        rgb_probes = f(N, n_probes, 3)
    with _clock('vqn_brdf_shade_fwd'):
        rc = lib().vqn_brdf_shade_fwd_rows(_ptr(rows), _ptr(xyz), _ptr(normal), _ptr(rayo), _ptr(lvis), _ptr(lxyz), _ptr(lareas),
                                           _ptr(light), ctypes.c_int64(N), ctypes.c_int(L), ctypes.c_int(len(materials)),
                                           *[_ptr(m) for m in mats], _ptr(gamma), _ptr(nout), _ptr(rgb[0]),
                                           _ptr(rgb[1] if len(rgb) > 1 else None), _ptr(rd), _ptr(rs), ctypes.c_int(int(raw)),
                                           _ptr(probes), ctypes.c_int(n_probes), _ptr(rgb_probes), _stream())
    _check(rc, 'vqn_brdf_shade_fwd_rows')
    return dict(rgb=rgb, normal=nout, rgb_diff=rd, rgb_spec=rs, rgb_probes=rgb_probes)


def neus_composite_bwd(rays_o, rays_d, mid_z, dists, sdf, grad, rgb, inv_s, background_rgb, radius, cos_anneal_ratio,
                       g_color, g_weight_sum=None, g_weights=None, g_gradient_error=None, gerr_den=None):
    """Reverse of neus_composite_fwd -> (g_sdf [B,n], g_grad [B,n,3], g_rgb [B,n,3], g_inv_s [B])."""
    for n_, t in (('rays_o', rays_o), ('rays_d', rays_d), ('mid_z', mid_z), ('dists', dists), ('sdf', sdf), ('grad', grad),
                  ('rgb', rgb), ('inv_s', inv_s), ('g_color', g_color)):
        _f32c(t, n_)
    B, n = mid_z.shape
    dev = mid_z.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    g_sdf, g_grad, g_rgb, g_inv_s = f(B, n), f(B, n, 3), f(B, n, 3), f(B)
    opt = lambda t, nm: None if t is None else _f32c(t.contiguous(), nm)
    g_weight_sum, g_weights = opt(g_weight_sum, 'g_weight_sum'), opt(g_weights, 'g_weights')
    g_gradient_error, gerr_den = opt(g_gradient_error, 'g_gradient_error'), opt(gerr_den, 'gerr_den')
    if background_rgb is not None:
        background_rgb = _f32c(background_rgb.reshape(-1)[:3].contiguous(), 'background_rgb')
    with _clock('vqn_neus_composite_bwd'):
        rc = lib().vqn_neus_composite_bwd(_ptr(rays_o), _ptr(rays_d), _ptr(mid_z), _ptr(dists), _ptr(sdf), _ptr(grad), _ptr(rgb),
                                          _ptr(inv_s), _ptr(background_rgb), ctypes.c_int64(B), ctypes.c_int(n),
                                          ctypes.c_float(float(radius)), ctypes.c_float(float(cos_anneal_ratio)), _ptr(g_color),
                                          _ptr(g_weight_sum), _ptr(g_weights), _ptr(g_gradient_error), _ptr(gerr_den),
                                          _ptr(g_sdf), _ptr(g_grad), _ptr(g_rgb), _ptr(g_inv_s), _stream())
    _check(rc, 'vqn_neus_composite_bwd')
    return g_sdf, g_grad, g_rgb, g_inv_s


def brdf_shade_bwd(xyz, normal, rayo, lvis, lxyz, lareas, light, materials, g_sums):
    """Reverse of brdf_shade_fwd(raw=True).  materials / g_sums: per set (albedo, spec, rough) and d loss / d sum [N,3].
    -> ([(g_albedo, g_spec, g_rough [N,1])...], g_light [L,3])."""
    N, L = xyz.shape[0], lareas.numel()
    dev = xyz.device
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    args_m, outs = [], []
    for (a, s, r), g in zip(materials, g_sums):
        for t in (a, s, r, g):
            _f32c(t, 'material')
        args_m += [a, s, r, g]
        outs.append((f(N, 3), f(N, 3), f(N, 1)))
    while len(args_m) < 8:
        args_m.append(None)
    flat_out = [t for o in outs for t in o]
    while len(flat_out) < 6:
        flat_out.append(None)
    L_ = lib()
    L_.vqn_brdf_shade_bwd_partials.restype = ctypes.c_int64
    n_part = int(L_.vqn_brdf_shade_bwd_partials(ctypes.c_int64(N)))
    part = f(n_part, L, 3)
    with _clock('vqn_brdf_shade_bwd'):
        rc = L_.vqn_brdf_shade_bwd(_ptr(xyz), _ptr(normal), _ptr(rayo), _ptr(lvis), _ptr(lxyz), _ptr(lareas), _ptr(light),
                                   ctypes.c_int64(N), ctypes.c_int(L), ctypes.c_int(len(materials)), *[_ptr(t) for t in args_m],
                                   *[_ptr(t) for t in flat_out], _ptr(part), _stream())
    _check(rc, 'vqn_brdf_shade_bwd')
    return outs, part.sum(0)


# -------------------------------------------------------------------------------------- reflectance training passes (round 4)
def refl_train_fwd_x3(desc, wbuf_pieces, wbuf_f32, pts, z_rows, P, saved, z_rows_out, head_out, split_heads=False, save=True, zx_rows=None,
                      zx_tiles_out=None):
    """Forward of a reflectance stack (optional encoder + up to three heads) on the exact-split engine, keeping what the backward
    needs (csrc/refl_train_x3.hip: vqn_refl_train_fwd_x3).  split_heads: one workgroup row per head (small batches).
    zx_rows [P, z]: the heads' second input (descriptors with zx_tiles > 0: vqn_refl_train_fwd_x3_zx); zx_tiles_out: its tile-format copy."""
    _f32c(wbuf_f32, 'wbuf_f32')
    for t in [t for t in saved if t is not None] + list(head_out) + [t for t in (pts, z_rows, z_rows_out, zx_rows, zx_tiles_out) if t is not None]:
        _f32c(t, 'tensor')
    d, dp = _i32(desc)
    sp = (ctypes.c_void_p * len(saved))(*[0 if t is None else t.data_ptr() for t in saved])
    hp = (ctypes.c_void_p * max(1, len(head_out)))(*[t.data_ptr() for t in head_out])
    with _clock('vqn_refl_train_fwd_x3'):
        if zx_rows is None:
            rc = lib().vqn_refl_train_fwd_x3(dp, _ptr(wbuf_pieces), _ptr(wbuf_f32), _ptr(pts), _ptr(z_rows), ctypes.c_int64(P), sp,
                                             ctypes.c_int(len(saved)), _ptr(z_rows_out), hp, ctypes.c_int(int(split_heads)), ctypes.c_int(int(save)),
                                             _stream())
        else:
            if zx_rows.shape[0] != P:
                raise VqnError('refl_train_fwd_x3: zx_rows must have one row per point')
            rc = lib().vqn_refl_train_fwd_x3_zx(dp, _ptr(wbuf_pieces), _ptr(wbuf_f32), _ptr(pts), _ptr(z_rows), _ptr(zx_rows), ctypes.c_int64(P), sp,
                                                ctypes.c_int(len(saved)), _ptr(z_rows_out), _ptr(zx_tiles_out), hp, ctypes.c_int(int(split_heads)),
                                                ctypes.c_int(int(save)), _stream())
    _check(rc, 'vqn_refl_train_fwd_x3')


def refl_train_bwd_x3(desc, wbuf_pieces, wbuf_f32, P, g_out, head_out, g_z_rows, saved, outs, gz_rows_out, run_heads=True, run_enc=True,
                      split_heads=False, d2_row0=None):
    """Backward of the same stack (vqn_refl_train_bwd_x3): fills `outs` with every layer's per-point adjoint in the tile format.
    g_z_rows: list of up to four [P, z] adjoints flowing into z from outside this launch's heads; run_heads / run_enc select the part
    of the stack walked; split_heads (heads only): gz_rows_out holds one [P, z] slice per head."""
    _f32c(wbuf_f32, 'wbuf_f32')
    g_z_rows = [t for t in (g_z_rows or []) if t is not None]
    for t in list(saved) + list(outs) + list(g_out) + list(head_out) + g_z_rows + [t for t in (gz_rows_out,) if t is not None]:
        _f32c(t, 'tensor')
    d, dp = _i32(desc)
    dev = wbuf_f32.device
    L = lib()
    L.vqn_refl_train_bwd_x3_scratch_bytes.restype = ctypes.c_int64
    need = int(L.vqn_refl_train_bwd_x3_scratch_bytes(dp))
    if need <= 0:
        raise VqnError('vqn_refl_train_bwd_x3_scratch_bytes: invalid descriptor')
    # one buffer per (device, stream, size), never replaced: a captured training step (Trainer(graph=True)) holds this pointer for as long
    # as its graph is replayed, so a later, larger request on the same stream must not free it
    key = (str(dev), torch.cuda.current_stream().cuda_stream, 'refl_bwd', need)
    buf = _scratch.get(key)
    if buf is None:
        buf = torch.empty((need,), dtype=torch.uint8, device=dev)
        _scratch[key] = buf
    arr = lambda ts: (ctypes.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])
    with _clock('vqn_refl_train_bwd_x3'):
        rc = L.vqn_refl_train_bwd_x3(dp, _ptr(wbuf_pieces), _ptr(wbuf_f32), ctypes.c_int64(P), arr(g_out), arr(head_out), arr(g_z_rows),
                                     ctypes.c_int(len(g_z_rows)), arr(saved), ctypes.c_int(len(saved)), arr(outs), ctypes.c_int(len(outs)),
                                     _ptr(gz_rows_out), None if d2_row0 is None else (ctypes.c_int32 * len(d2_row0))(*[int(v) for v in d2_row0]),
                                     ctypes.c_int(int(run_heads)), ctypes.c_int(int(run_enc)), ctypes.c_int(int(split_heads)),
                                     _ptr(buf), ctypes.c_int64(buf.numel()), _stream())
    _check(rc, 'vqn_refl_train_bwd_x3')


# -------------------------------------------------------------------------------------- element-wise pieces (round 4)
def clip_preserve(x, lo, hi):
    """x + (clip(x) - x) in one launch (vqn_clip_preserve)."""
    _f32c(x, 'x')
    y = torch.empty_like(x)
    with _clock('vqn_clip_preserve'):
        rc = lib().vqn_clip_preserve(_ptr(x), ctypes.c_int64(x.numel()), ctypes.c_float(lo), ctypes.c_float(hi), _ptr(y), _stream())
    _check(rc, 'vqn_clip_preserve')
    return y


def ks_split_fwd(basecolor, ks):
    _f32c(basecolor, 'basecolor'); _f32c(ks, 'ks')
    albedo, spec = torch.empty_like(basecolor), torch.empty_like(basecolor)
    with _clock('vqn_ks_split_fwd'):
        rc = lib().vqn_ks_split_fwd(_ptr(basecolor), _ptr(ks), ctypes.c_int(ks.shape[1]), ctypes.c_int64(basecolor.shape[0]), _ptr(albedo),
                                    _ptr(spec), _stream())
    _check(rc, 'vqn_ks_split_fwd')
    return albedo, spec


def ks_split_bwd(basecolor, ks, g_albedo, g_spec):
    for t in (g_albedo, g_spec):
        if t is not None:
            _f32c(t, 'gradient')
    g_bc, g_ks = torch.empty_like(basecolor), torch.empty_like(ks)
    with _clock('vqn_ks_split_bwd'):
        rc = lib().vqn_ks_split_bwd(_ptr(basecolor), _ptr(ks), ctypes.c_int(ks.shape[1]), ctypes.c_int64(basecolor.shape[0]), _ptr(g_albedo),
                                    _ptr(g_spec), _ptr(g_bc), _ptr(g_ks), _stream())
    _check(rc, 'vqn_ks_split_bwd')
    return g_bc, g_ks


def loss_total(terms, vqloss, sim, use_chr, use_smooth, use_lambert):
    _f32c(terms, 'terms'); _f32c(vqloss, 'vqloss')
    if sim is not None:
        _f32c(sim, 'sim')
    out = torch.empty((terms.shape[0],), dtype=torch.float32, device=terms.device)
    with _clock('vqn_loss_total'):
        rc = lib().vqn_loss_total(_ptr(terms), ctypes.c_int64(terms.shape[0]), _ptr(vqloss), _ptr(sim), ctypes.c_int(int(use_chr)),
                                  ctypes.c_int(int(use_smooth)), ctypes.c_int(int(use_lambert)), _ptr(out), _stream())
    _check(rc, 'vqn_loss_total')
    return out
