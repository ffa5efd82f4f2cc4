"""ctypes binding of oracle/vq_strict.c (test infrastructure; see that file's header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libvq_strict.so')
_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, 'vq_strict.c')):
        subprocess.check_call(['make', '-C', _HERE, '-s', '-B'])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.vq_strict_assign.restype = ctypes.c_int
        _lib.vq_strict_ema_stats.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def assign(x, C, sel=None, want_dist=True, want_quant=True):
    x = np.ascontiguousarray(x, np.float32)
    C = np.ascontiguousarray(C, np.float32)
    N, D = x.shape
    K = C.shape[1]
    sel = None if sel is None else np.ascontiguousarray(sel, np.float32).reshape(K)
    dist = np.empty((N, K), np.float32) if want_dist else None
    quant = np.empty((N, D), np.float32) if want_quant else None
    idx = np.empty((N,), np.int64)
    rc = lib().vq_strict_assign(_p(x), ctypes.c_long(N), ctypes.c_int(D), _p(C), ctypes.c_int(K), _p(sel),
                                _p(dist), _p(idx), _p(quant))
    assert rc == 0, rc
    return idx, dist, quant


def ema_stats(x, idx, K):
    x = np.ascontiguousarray(x, np.float32)
    idx = np.ascontiguousarray(idx, np.int64)
    N, D = x.shape
    counts = np.empty((K,), np.float32)
    dw = np.empty((D, K), np.float32)
    rc = lib().vq_strict_ema_stats(_p(x), _p(idx), ctypes.c_long(N), ctypes.c_int(D), ctypes.c_int(K), _p(counts), _p(dw))
    assert rc == 0, rc
    return counts, dw


def l2_normalize(x, eps=1e-6):
    """vq_strict_l2_normalize: rows of x scaled to unit length in the pinned order (see vq_strict.c)."""
    x = np.ascontiguousarray(x, np.float32)
    N, D = x.shape
    y = np.empty_like(x)
    rc = lib().vq_strict_l2_normalize(_p(x), ctypes.c_long(N), ctypes.c_int(D), ctypes.c_float(eps), _p(y))
    assert rc == 0, rc
    return y
