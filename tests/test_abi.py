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
    # round-3 entry points: the same contract (bad argument -> -1 with a message, empty work -> 0), before anything touches a device
    assert lib.vqn_neus_train_fwd(None, None, None, None, None, None, ctypes.c_int64(8), None, ctypes.c_int64(0), None, 0, 2, 9, 2, None, None,
                                  None, None) == -1 and b'null pointer' in lib.vqn_last_error()
    assert lib.vqn_neus_train_bwd(None, None, None, None, None, None, None, ctypes.c_int64(8), None, ctypes.c_int64(0), None, 0, None, 0,
                                  None) == -1
    lib.vqn_neus_train_bwd_scratch_bytes.restype = ctypes.c_int64
    assert lib.vqn_neus_train_bwd_scratch_bytes(None) == -1
    bad = (ctypes.c_int32 * 76)()                                   # all zeros: no layers
    assert lib.vqn_neus_train_bwd_scratch_bytes(bad) == -1
    assert lib.vqn_wgrad_finalize(1, None, None, None, None, None, None, None, None, None, None, None, None, None, None) == -1
    assert lib.vqn_multi_copy(1, None, None, None, None) == -1
    assert lib.vqn_multi_copy(0, (ctypes.c_void_p * 1)(), (ctypes.c_void_p * 1)(), (ctypes.c_int64 * 1)(), None) == 0
    assert lib.vqn_wgrad_partials_batched(1, None, None, None, None, None, None, None, None, ctypes.c_int64(4), 256, None, None, 1, None) == -1
    assert lib.vqn_l2_normalize_rows_bwd(None, None, ctypes.c_int64(0), 256, ctypes.c_float(1e-6), None, None) == 0
    assert lib.vqn_l2_normalize_rows_bwd(None, None, ctypes.c_int64(4), 255, ctypes.c_float(1e-6), None, None) == -1
    assert lib.vqn_vq_ste_loss_bwd(None, None, None, None, ctypes.c_int64(8), None, None) == -1
    assert lib.vqn_tfmt_pack_delta(None, ctypes.c_int64(4), 3, ctypes.c_int64(3), None, 2, None, 1, None) == -1      # softplus is not a head activation


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
    for piece in (0, 1, 2):                                              # kind 3 / 4 / 5: exact bf16x3 split by truncation
        m = kind == 3 + piece
        if not m.any():
            continue
        halves = []
        for idx in (i0[m], i1[m]):
            r = get(src[m], idx)
            for _ in range(piece + 1):
                p = (r.view(np.uint32) & np.uint32(0xffff0000)).view(np.float32)
                r = r - p
            halves.append((p.view(np.uint32) >> np.uint32(16)).astype(np.uint16))
        out[m] = np.stack(halves, -1).copy().view(np.float32)[:, 0]
    return out


SDF_SHAPES = [
    # dims, skip_in, multires
    ([39, 256, 256, 256, 256, 256, 256, 256, 256, 257], (4,), 6),        # nerf.conf
    ([39, 64, 64, 65], (), 6),                                          # BASELINE configs[0]
    ([27, 100, 100, 100, 33], (2,), 4),
    ([39, 48, 48, 48, 48, 48, 1], (3,), 6),                             # no feature rows (d_out = 1)
]


@pytest.mark.parametrize('f16s', [0, 1, 2])              # engine mode of the packs: 0 f32, 1 f16 pair, 2 bf16x3 (exact split)
@pytest.mark.parametrize('dims,skip_in,multires', SDF_SHAPES)
def test_c_sdf_pack_equals_python_pack(dims, skip_in, multires, f16s):
    from vqnerf_release_amd.geo import packing
    lib = _C.lib()
    lib.vqn_neus_sdf_pack_plan.restype = ctypes.c_int64
    mode = ('f32', 'f16s', 'x3')[f16s]
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
        if f16s == 1:
            g16, w16 = got[scaled].view(np.float16).astype(np.float32), want[scaled].view(np.float16).astype(np.float32)
            hi = (words[scaled, 3] == 1).repeat(2)
            np.testing.assert_allclose(g16[hi], w16[hi], rtol=2e-3, atol=0)
        elif f16s == 2:
            # leading pieces (8 significant bits of a value that differs by one f32 ulp): equal or one bf16 ulp apart
            gb = (got[scaled].view(np.uint16).astype(np.uint32) << 16).view(np.float32)
            wb = (want[scaled].view(np.uint16).astype(np.uint32) << 16).view(np.float32)
            p0 = (words[scaled, 3] == 3).repeat(2)
            np.testing.assert_allclose(gb[p0], wb[p0], rtol=2 ** -7, atol=0)
        else:
            np.testing.assert_allclose(got[scaled], want[scaled], rtol=2.5e-7, atol=0)
        assert skip > 0


@pytest.mark.parametrize('f16s', [0, 1, 2])
@pytest.mark.parametrize('d_feature,mode,d_hidden,n_layers,mv', [(256, 'idr', 256, 4, 4), (64, 'idr', 64, 2, 4), (32, 'no_view_dir', 100, 3, 0),
                                                                  (96, 'no_normal', 48, 2, 2)])
def test_c_colour_pack_equals_python_pack(d_feature, mode, d_hidden, n_layers, mv, f16s):
    from vqnerf_release_amd.geo import packing
    lib = _C.lib()
    lib.vqn_neus_col_pack_plan.restype = ctypes.c_int64
    ft = (d_feature + 31) // 32
    plan = packing.ColPackPlan(d_feature, mode, d_hidden, n_layers, 3, mv, True, ft, matrix_mode=('f32', 'f16s', 'x3')[f16s])
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


# ---- vqn_chain_pack_plan: the C layer-program builder against the Python one (decomp/packing.py), host-only -------------------
class _Stack(ctypes.Structure):
    _fields_ = [('kind', ctypes.c_int32), ('n_layers', ctypes.c_int32), ('widths', ctypes.c_int32 * 8), ('acts', ctypes.c_int32 * 8),
                ('skip_at', ctypes.c_int32), ('input', ctypes.c_int32), ('out_slot', ctypes.c_int32)]


def _stack(kind, widths, acts, skip_at=-1, input=-1, out_slot=-1):
    from vqnerf_release_amd.decomp.packing import ACT
    s = _Stack()
    s.kind, s.n_layers, s.skip_at, s.input, s.out_slot = kind, len(widths), skip_at, input, out_slot
    for i, (w, a) in enumerate(zip(widths, acts)):
        s.widths[i], s.acts[i] = w, ACT[a]
    return s


def _chain_model(mlp_width=128, z=256, nf=10):
    from tests.decomp_util import make_config
    from vqnerf_release_amd.decomp.nerfactor.models import get_model_class
    m = get_model_class('vq_nfr')(make_config(mlp_width=mlp_width, conv_width=z, n_freqs_xyz=nf))
    m.build_nets(device='cpu', seed=3)
    with torch.no_grad():
        for net in m.net.values():
            for layer in net.layers:
                layer.bias.uniform_(-0.1, 0.1)
    return m


@pytest.mark.parametrize('program', ['enc', 'heads', 'enc+heads', 'one_head'])
@pytest.mark.parametrize('shape', [(128, 256, 10), (64, 128, 6), (96, 160, 4)])
def test_c_chain_program_equals_python_program(program, shape):
    lib = _C.lib()
    lib.vqn_chain_pack_plan.restype = ctypes.c_int64
    lib.vqn_last_error.restype = ctypes.c_char_p
    w, z, nf = shape
    m = _chain_model(*shape)
    fe, bn = m.net['fine_enc'], m.net['bottleneck']
    names = ['diff_main', 'spec_main', 'rough_main'] if program != 'one_head' else ['rough_vq']
    enc_stacks = [_stack(0, fe.widths, fe.act, skip_at=fe.skip_at[0]), _stack(0, bn.widths, bn.act, input=0, out_slot=0)]
    head = lambda n, inp, slot: _stack(1, m.net[n].widths, m.net[n].act, skip_at=1, input=inp, out_slot=slot)
    if program == 'enc':
        plan, nets, stacks, in_mode, in_feats = m._enc_program(), ['fine_enc', 'bottleneck'], enc_stacks, 1, 3 + 6 * nf
    elif program == 'enc+heads':
        plan, nets = m._enc_heads_program(names), ['fine_enc', 'bottleneck'] + names
        stacks, in_mode, in_feats = enc_stacks + [head(n, 1, i + 1) for i, n in enumerate(names)], 1, 3 + 6 * nf
    else:
        plan, nets = m._head_program(names), names
        stacks, in_mode, in_feats = [head(n, -1, i) for i, n in enumerate(names)], 0, z
    params, kernels, biases = {}, [], []
    for n in nets:
        for i, layer in enumerate(m.net[n].layers):
            params[f'{n}/{i}'] = (layer.kernel.detach(), layer.bias.detach())
            kernels.append(layer.kernel.detach().numpy().reshape(-1)); biases.append(layer.bias.detach().numpy())
    want_wbuf, want_desc = plan.pack(params)
    arr = (_Stack * len(stacks))(*stacks)
    desc = np.zeros(16 + 16 * 16, np.int32)
    n = lib.vqn_chain_pack_plan(in_mode, in_feats, nf if in_mode else 0, len(stacks), arr, desc.ctypes.data_as(ctypes.c_void_p), None, ctypes.c_int64(0))
    assert n == want_wbuf.numel(), (n, want_wbuf.numel(), lib.vqn_last_error())
    # the <= 4-output layers carry their biases in the descriptor: the plan leaves them to vqn_chain_pack_update
    for li in range(int(desc[0])):
        base = 16 + 16 * li
        if desc[base] == 1:
            desc[base + 12: base + 16] = want_desc[base + 12: base + 16]
    np.testing.assert_array_equal(desc, want_desc)
    words = np.zeros((n, 4), np.int32)
    lib.vqn_chain_pack_plan(in_mode, in_feats, nf if in_mode else 0, len(stacks), arr, None, words.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n))
    arrays = {}
    for i, (k, b) in enumerate(zip(kernels, biases)):
        arrays[2 * i], arrays[2 * i + 1] = k, b
    np.testing.assert_array_equal(_apply_words(words, arrays).view(np.int32), want_wbuf.numpy().view(np.int32))


def test_c_chain_program_rejects_what_it_does_not_build():
    lib = _C.lib()
    lib.vqn_chain_pack_plan.restype = ctypes.c_int64
    bad_head = (_Stack * 1)(_stack(1, [256, 256, 3], ['relu', 'relu', 'sigmoid'], skip_at=1, out_slot=0))      # middle layer > 128 wide
    assert lib.vqn_chain_pack_plan(0, 256, 0, 1, bad_head, None, None, ctypes.c_int64(0)) == -2
    fwd_ref = (_Stack * 1)(_stack(0, [64], ['relu'], input=0))                                               # reads a stack that is not there yet
    assert lib.vqn_chain_pack_plan(0, 64, 0, 1, fwd_ref, None, None, ctypes.c_int64(0)) == -1
    assert lib.vqn_chain_pack_plan(1, 60, 10, 1, (_Stack * 1)(_stack(0, [64], ['relu'])), None, None, ctypes.c_int64(0)) == -2   # 60 != 3 + 6 * 10
