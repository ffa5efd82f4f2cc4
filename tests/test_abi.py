"""CPU: the C-ABI library builds, loads, and exports every symbol include/vqnerf_hip.h declares."""
import ctypes
import os
import re

from vqnerf_release_amd import _C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'vqnerf_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(vqn_[a-z0-9_]+)\s*\(', txt)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_C.LIB_PATH):
        _C.build()
    lib = ctypes.CDLL(_C.LIB_PATH)
    names = _declared()
    assert len(names) >= 4
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert _C.lib().vqn_version() >= 100


def test_argument_errors_do_not_need_a_gpu():
    lib = _C.lib()
    lib.vqn_last_error.restype = ctypes.c_char_p
    rc = lib.vqn_vq_assign(None, ctypes.c_int64(-1), 256, None, 16, None, None, None, None, None, None)
    assert rc == -1 and b'bad argument' in lib.vqn_last_error()
    rc = lib.vqn_vq_assign(None, ctypes.c_int64(0), 256, None, 16, None, None, None, None, None, None)
    assert rc == 0          # empty input is a no-op


# ---- vqn_neus_*_pack_plan: the C pack builder against the Python one (host-only halves, no GPU needed) ------------------------
import numpy as np
import pytest
import torch


def _apply_words(words, arrays):
    """numpy statement of pack_gather_kernel (csrc/neus_pack.hip): words [n,4] = (src, i0, i1, kind); arrays[2l] = W_l flat,
    arrays[2l+1] = bias_l.  The skip layer's 1/sqrt(2) is a multiplication by the f32 reciprocal (see the kernel)."""
    n = len(words)
    out = np.zeros(n, np.float32)
    inv = np.float32(1.0) / np.float32(np.sqrt(2.0))

    def get(src, idx):
        v = np.zeros(len(idx), np.float32)
        for s in np.unique(src):
            if s < 0:
                continue
            m = (src == s) & (idx >= 0)
            a = arrays[int(s) & 0xff]
            vals = a[idx[m]]
            if int(s) & (1 << 30):
                vals = vals * inv
            v[m] = vals
        return v
    src, i0, i1, kind = words[:, 0], words[:, 1], words[:, 2], words[:, 3]
    k0 = kind == 0
    out[k0] = get(src[k0], i0[k0])
    for hl in (1, 2):
        m = kind == hl
        if not m.any():
            continue
        halves = []
        for idx in (i0[m], i1[m]):
            v = get(src[m], idx)
            hi = v.astype(np.float16)
            halves.append(hi if hl == 1 else ((v - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16))
        pair = np.stack(halves, -1).copy()                               # low half first
        out[m] = pair.view(np.float32)[:, 0]
    return out


SDF_SHAPES = [
    # dims, skip_in, multires
    ([39, 256, 256, 256, 256, 256, 256, 256, 256, 257], (4,), 6),        # nerf.conf
    ([39, 64, 64, 65], (), 6),                                          # BASELINE configs[0]
    ([27, 100, 100, 100, 33], (2,), 4),
    ([39, 48, 48, 48, 48, 48, 1], (3,), 6),                             # no feature rows (d_out = 1)
]


@pytest.mark.parametrize('f16s', [0, 1])
@pytest.mark.parametrize('dims,skip_in,multires', SDF_SHAPES)
def test_c_sdf_pack_equals_python_pack(dims, skip_in, multires, f16s):
    from vqnerf_release_amd.geo import packing
    lib = _C.lib()
    lib.vqn_neus_sdf_pack_plan.restype = ctypes.c_int64
    mode = 'f16s' if f16s else 'f32'
    plan = packing.SdfPackPlan(dims, skip_in, multires, 1.5, max_tiles=8, mode=mode)
    n_lin = len(dims) - 1
    rng = np.random.default_rng(len(dims) + multires)
    W = [torch.tensor(rng.normal(size=(plan.out_dims[l], plan.in_dims[l])).astype(np.float32)) for l in range(n_lin)]
    b = [torch.tensor(rng.normal(size=(plan.out_dims[l],)).astype(np.float32)) for l in range(n_lin)]
    want_wbuf, want_desc = plan.pack(W, b)                                # torch CPU: a true division by sqrt(2) in the skip layer
    cdims = (ctypes.c_int32 * len(dims))(*dims)
    desc = np.zeros(packing.SDF_DESC_INTS, np.int32)
    skip = plan.skip
    n = lib.vqn_neus_sdf_pack_plan(cdims, n_lin, skip, multires, ctypes.c_float(1.5), 8, 1, f16s, desc.ctypes.data_as(ctypes.c_void_p), None,
                                   ctypes.c_int64(0))
    assert n == want_wbuf.numel()
    np.testing.assert_array_equal(desc, want_desc)
    words = np.zeros((n, 4), np.int32)
    assert lib.vqn_neus_sdf_pack_plan(cdims, n_lin, skip, multires, ctypes.c_float(1.5), 8, 1, f16s, None,
                                      words.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n)) == n
    arrays = {}
    for l in range(n_lin):
        arrays[2 * l], arrays[2 * l + 1] = W[l].numpy().reshape(-1), b[l].numpy()
    got = _apply_words(words, arrays)
    want = want_wbuf.numpy()
    scaled = (words[:, 0] >= 0) & ((words[:, 0] & (1 << 30)) != 0)
    np.testing.assert_array_equal(got[~scaled].view(np.int32), want[~scaled].view(np.int32))     # bit for bit
    if scaled.any():
        # x * (1 / sqrt 2) (device arithmetic of the Python pack, and of the C kernel) vs x / sqrt 2 (torch on the CPU): one ulp
        if f16s:
            g16, w16 = got[scaled].view(np.float16).astype(np.float32), want[scaled].view(np.float16).astype(np.float32)
            hi = (words[scaled, 3] == 1).repeat(2)
            np.testing.assert_allclose(g16[hi], w16[hi], rtol=2e-3, atol=0)
        else:
            np.testing.assert_allclose(got[scaled], want[scaled], rtol=2.5e-7, atol=0)
        assert skip > 0


@pytest.mark.parametrize('f16s', [0, 1])
@pytest.mark.parametrize('d_feature,mode,d_hidden,n_layers,mv', [(256, 'idr', 256, 4, 4), (64, 'idr', 64, 2, 4), (32, 'no_view_dir', 100, 3, 0),
                                                                  (96, 'no_normal', 48, 2, 2)])
def test_c_colour_pack_equals_python_pack(d_feature, mode, d_hidden, n_layers, mv, f16s):
    from vqnerf_release_amd.geo import packing
    lib = _C.lib()
    lib.vqn_neus_col_pack_plan.restype = ctypes.c_int64
    ft = (d_feature + 31) // 32
    plan = packing.ColPackPlan(d_feature, mode, d_hidden, n_layers, 3, mv, True, ft, matrix_mode='f16s' if f16s else 'f32')
    rng = np.random.default_rng(d_feature + n_layers)
    W = [torch.tensor(rng.normal(size=(plan.dims[l + 1], plan.dims[l])).astype(np.float32)) for l in range(plan.n_lin)]
    b = [torch.tensor(rng.normal(size=(plan.dims[l + 1],)).astype(np.float32)) for l in range(plan.n_lin)]
    want_wbuf, want_desc = plan.pack(W, b)
    cmode = {'idr': 0, 'no_view_dir': 1, 'no_normal': 2}[mode]
    desc = np.zeros(packing.COL_DESC_INTS, np.int32)
    n = lib.vqn_neus_col_pack_plan(d_feature, cmode, d_hidden, n_layers, 3, mv, 1, ft, f16s, desc.ctypes.data_as(ctypes.c_void_p), None,
                                   ctypes.c_int64(0))
    assert n == want_wbuf.numel()
    np.testing.assert_array_equal(desc, want_desc)
    words = np.zeros((n, 4), np.int32)
    lib.vqn_neus_col_pack_plan(d_feature, cmode, d_hidden, n_layers, 3, mv, 1, ft, f16s, None, words.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_int64(n))
    arrays = {}
    for l in range(plan.n_lin):
        arrays[2 * l], arrays[2 * l + 1] = W[l].numpy().reshape(-1), b[l].numpy()
    np.testing.assert_array_equal(_apply_words(words, arrays).view(np.int32), want_wbuf.numpy().view(np.int32))


def test_c_pack_plan_rejects_unsupported_shapes():
    lib = _C.lib()
    lib.vqn_neus_sdf_pack_plan.restype = ctypes.c_int64
    lib.vqn_last_error.restype = ctypes.c_char_p
    dims = (ctypes.c_int32 * 4)(39, 64, 64, 65)
    assert lib.vqn_neus_sdf_pack_plan(dims, 3, 2, 6, ctypes.c_float(1.0), 0, 1, 0, None, None, ctypes.c_int64(0)) == -2   # skip into the last layer
    assert b'skip' in lib.vqn_last_error()
    bad = (ctypes.c_int32 * 4)(40, 64, 64, 65)
    assert lib.vqn_neus_sdf_pack_plan(bad, 3, -1, 6, ctypes.c_float(1.0), 0, 1, 0, None, None, ctypes.c_int64(0)) == -2
