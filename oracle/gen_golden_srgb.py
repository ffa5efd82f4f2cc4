#!/usr/bin/env python
"""Generate tests/golden/srgb.npz by RUNNING THE REAL REFERENCE's numpy branches (decomp half, SURVEY 8 row a19 + the image helpers of f2).

Build-container only.  /root/reference/decomp/nerfvq_nfr3/nerfactor/util/img.py guards its TensorFlow import (`try: import tensorflow`,
img.py:20-25) and every colour helper has a numpy branch chosen by `isinstance(x, tf.Tensor)` (img.py:62-75, 78-97, 142-186, 189-201).
To import it here two NAMES must resolve and nothing else: `tensorflow.Tensor` (a type no numpy array is an instance of, so the numpy
branch runs) and `absl.logging` (verbosity constant + set_verbosity for util/logging.py:15-18).  No arithmetic is stood in for:
np.power / np.where / np.clip are the reference's own calls.  Only outputs (and the edge-value inputs) are stored; the reference source
never ships.  This is the one piece of the decomp half the real reference can pin (missing #8).

    python oracle/gen_golden_srgb.py            # writes tests/golden/srgb.npz
"""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('VQNERF_REFERENCE', '/root/reference')
GOLD = os.path.join(ROOT, 'tests', 'golden')


def import_reference_img():
    if 'tensorflow' not in sys.modules:
        tf = types.ModuleType('tensorflow')
        tf.Tensor = type('Tensor', (), {})                      # nothing is an instance of it: the numpy branches run
        sys.modules['tensorflow'] = tf
    if 'absl' not in sys.modules:
        absl, lg = types.ModuleType('absl'), types.ModuleType('absl.logging')
        lg.INFO = 0
        lg.set_verbosity = lambda *_a, **_k: None
        absl.logging = lg
        sys.modules['absl'], sys.modules['absl.logging'] = absl, lg
    sys.path.insert(0, os.path.join(REF, 'decomp', 'nerfvq_nfr3'))
    from nerfactor.util import img                              # noqa: E402
    return img


def edge_values():
    """fp32 inputs: <= 0, the two thresholds +- 1 and 2 ulp, tiny / denormal, 1 -+ ulp, > 1, a dense ramp and seeded uniforms"""
    f = np.float32
    up = lambda v, n=1: np.nextafter(f(v), f(np.inf), dtype=f) if n == 1 else np.nextafter(np.nextafter(f(v), f(np.inf), dtype=f), f(np.inf), dtype=f)
    dn = lambda v, n=1: np.nextafter(f(v), f(-np.inf), dtype=f) if n == 1 else np.nextafter(np.nextafter(f(v), f(-np.inf), dtype=f), f(-np.inf), dtype=f)
    pts = [f(-1.0), f(-1e-3), f(-0.0), f(0.0), f(1e-45), f(1e-38), f(1e-20), f(1e-6)]
    for t in (0.0031308, 0.04045):
        pts += [dn(t, 2), dn(t), f(t), up(t), up(t, 2)]
    pts += [f(0.5), dn(1.0), f(1.0), up(1.0), f(1.5), f(2.0), f(100.0)]
    ramp = np.linspace(0.0, 1.0, 2049, dtype=f)
    uni = np.random.default_rng(19).uniform(-0.25, 1.25, 4096).astype(f)
    return np.concatenate([np.asarray(pts, dtype=f), ramp, uni])


def main():
    img = import_reference_img()
    x = edge_values()
    out = {'x': x}
    out['linear2srgb'] = img.linear2srgb(x.copy())                 # img.py:142-165 (clips to [0, 1] first, :155)
    out['srgb2linear'] = img.srgb2linear(x.copy())                 # img.py:167-186 (no clip; negative base -> NaN in the unused branch)
    out['clip_0to1'] = img._clip_0to1_warn(x.copy())               # img.py:62-75
    out['to_uint8'] = img.to_uint(x.copy(), 'uint8')               # img.py:189-201 (numpy branch: truncating cast of 255 * clip(x))
    out['to_uint16'] = img.to_uint(x.copy(), 'uint16')
    with np.errstate(invalid='ignore'):
        xn = np.concatenate([x, np.asarray([np.nan], np.float32)])
        out['x_with_nan'] = xn
        out['srgb2linear_with_nan'] = img.srgb2linear(xn.copy())
    # fp64 input (what np.float64 callers get): the dtype follows the input
    out['linear2srgb_f64'] = img.linear2srgb(x.astype(np.float64))
    out['srgb2linear_f64'] = img.srgb2linear(x.astype(np.float64))
    # alpha_blend numpy branch (img.py:78-97): HxWx3 with a HxW alpha, with and without a second image
    rng = np.random.default_rng(23)
    a = rng.uniform(0, 1, (5, 7, 3)).astype(np.float32)
    b = rng.uniform(0, 1, (5, 7, 3)).astype(np.float32)
    al = rng.uniform(0, 1, (5, 7)).astype(np.float32)
    out['blend_a'], out['blend_b'], out['blend_alpha'] = a, b, al
    out['alpha_blend_zero_bg'] = img.alpha_blend(a, al)
    out['alpha_blend_two'] = img.alpha_blend(a, al, b)
    for k in ('linear2srgb', 'srgb2linear'):
        assert out[k].dtype == np.float32, (k, out[k].dtype)
    os.makedirs(GOLD, exist_ok=True)
    np.savez_compressed(os.path.join(GOLD, 'srgb.npz'), **out)
    print('wrote', os.path.join(GOLD, 'srgb.npz'), {k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == '__main__':
    main()
