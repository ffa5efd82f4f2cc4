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
