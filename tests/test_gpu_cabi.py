"""GPU: the C ABI is self-sufficient -- a render whose weight packs and descriptors are built by vqn_neus_pack_* (C), not by the
Python packers, compared with the reference's golden outputs; and the C packs equal the Python packs bit for bit on the device."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('flag', ['', '--f16s', '--x3'])
def test_render_through_the_c_abi_alone(flag):
    """tests/cabi_render.py in a child process: ctypes + torch-as-allocator only, no module of the package imported."""
    cmd = [sys.executable, os.path.join(ROOT, 'tests', 'cabi_render.py')] + ([flag] if flag else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'imported package modules: []' in r.stdout
    print(r.stdout.strip())


def test_reflectance_call_through_the_c_abi_alone():
    """tests/cabi_reflectance.py in a child process: layer programs + packs from vqn_chain_pack_*, then encoder + heads, the fused
    quantiser, the VQ heads and the shading kernel -- against oracle/decomp.py, no module of the package imported."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'cabi_reflectance.py')], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'imported package modules: []' in r.stdout
    print(r.stdout.strip())


@pytest.mark.parametrize('mode', ['f32', 'f16s', 'x3'])
@pytest.mark.parametrize('name', ['full', 'small'])
def test_c_packs_equal_python_packs_on_the_device(name, mode):
    from tests.test_gpu_neus_render import _build
    from vqnerf_release_amd import _C
    cfg, sdf, col, var, ren = _build(name)
    ren.matrix_mode = mode
    wb_s, d_s, wb_c, d_c = ren._packs()
    lib = _C.lib()
    for f in ('vqn_neus_pack_sdf_desc', 'vqn_neus_pack_col_desc', 'vqn_neus_pack_sdf_wbuf', 'vqn_neus_pack_col_wbuf'):
        getattr(lib, f).restype = ctypes.c_void_p
    lib.vqn_neus_pack_sdf_floats.restype = lib.vqn_neus_pack_col_floats.restype = ctypes.c_int64
    dims = list(sdf.dims)
    s_lins = [getattr(sdf, f'lin{l}') for l in range(sdf.num_layers - 1)]
    c_lins = [getattr(col, f'lin{l}') for l in range(col.num_layers - 1)]
    with torch.no_grad():
        W = [m.effective_weight().float().contiguous() for m in s_lins + c_lins]
        b = [m.bias.detach().float().contiguous() for m in s_lins + c_lins]
    arr = lambda ts: (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    pack = ctypes.c_void_p()
    skip = [l for l in sdf.skip_in if 0 < l < len(dims) - 1]
    rc = lib.vqn_neus_pack_create((ctypes.c_int32 * len(dims))(*dims), len(dims) - 1, skip[0] if skip else -1, sdf.multires,
                                  ctypes.c_float(sdf.scale), 0, col.dims[1], col.num_layers - 2, col.multires_view, int(col.squeeze_out),
                                  {'f32': 0, 'f16s': 1, 'x3': 2}[mode], ctypes.byref(pack))
    assert rc == 0, lib.vqn_last_error()
    try:
        ns = len(s_lins)
        rc = lib.vqn_neus_pack_update(pack, arr(W[:ns]), arr(b[:ns]), arr(W[ns:]), arr(b[ns:]),
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.vqn_last_error()
        torch.cuda.synchronize()
        for got_n, got_p, got_d, want_w, want_d in (
                (lib.vqn_neus_pack_sdf_floats(pack), lib.vqn_neus_pack_sdf_wbuf(pack), lib.vqn_neus_pack_sdf_desc(pack), wb_s, d_s),
                (lib.vqn_neus_pack_col_floats(pack), lib.vqn_neus_pack_col_wbuf(pack), lib.vqn_neus_pack_col_desc(pack), wb_c, d_c)):
            assert got_n == want_w.numel()
            desc = np.ctypeslib.as_array(ctypes.cast(got_d, ctypes.POINTER(ctypes.c_int32)), shape=(len(want_d),))
            np.testing.assert_array_equal(desc, want_d)
            host = torch.empty(got_n, dtype=torch.float32)
            torch.cuda.synchronize()
            # device -> host copy of the C pack through a torch view of the raw pointer
            import ctypes as C
            hip = C.CDLL('libamdhip64.so')
            assert hip.hipMemcpy(C.c_void_p(host.data_ptr()), C.c_void_p(got_p), C.c_size_t(got_n * 4), 2) == 0
            assert torch.equal(host.view(torch.int32), want_w.cpu().view(torch.int32))          # bit for bit
    finally:
        lib.vqn_neus_pack_destroy(pack)
