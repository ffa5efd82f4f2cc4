"""The sRGB transfer curves + image helpers against outputs of the REAL reference (decomp half).

tests/golden/srgb.npz holds what /root/reference/decomp/nerfvq_nfr3/nerfactor/util/img.py's numpy branches returned for the edge values
oracle/gen_golden_srgb.py builds (<= 0, both thresholds +- 1 / 2 ulp, denormals, 1 -+ ulp, > 1, NaN, a ramp, seeded uniforms): the one
piece of the TensorFlow half the reference itself can pin here (VERDICT r04 missing #8).  Checked against it:
the oracle's restatement (oracle/decomp.py), the product's torch statement (util/img.py, the autograd path), the output-path helpers
(util/vis.py) and -- on the GPU -- the fused kernel vqn_linear2srgb.  Tolerances (numpy's, torch's and the device's fp32 `pow` differ in
the last place): linear2srgb 2.4e-7 absolute (2 ulp of its [0, 1] range: the curve subtracts 0.055 after the power), srgb2linear 4 ulp
of the result; rows on the linear branches (one fp32 multiply / divide) and the branch choice at the thresholds exact."""
import os

import numpy as np
import pytest
import torch

from oracle import decomp as od
from vqnerf_release_amd.decomp.nerfactor.util import img as imgutil
from vqnerf_release_amd.decomp.nerfactor.util import vis as visutil

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'srgb.npz'))


def _ulp_diff(a, b):
    """distance in units of the last place of float32 `b` (NaN == NaN counts as 0)"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    both_nan = np.isnan(a) & np.isnan(b)
    sp = np.spacing(np.maximum(np.abs(b), np.float32(1e-30))).astype(np.float64)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)) / sp
    d[both_nan] = 0.0
    assert not (np.isnan(a) ^ np.isnan(b)).any(), 'NaN pattern differs'
    return d


L2S_ATOL = 2.4e-7           # observed against the reference outputs: 1.2e-7 (oracle, torch statement)
S2L_ULPS = 4.0              # observed: 2.1 ulp


def _l2s_close(got, want):
    return float(np.abs(np.asarray(got, np.float64) - np.asarray(want, np.float64)).max()) <= L2S_ATOL


def _linear_branch_exact(x, got, thres, fwd):
    """rows the reference sends down the linear branch are a single fp32 multiply / divide: they must agree to the bit"""
    lin = np.clip(x, 0, 1) <= np.float32(thres) if fwd else x <= np.float32(thres)
    want = G['linear2srgb' if fwd else 'srgb2linear']
    assert lin.sum() > 10
    np.testing.assert_array_equal(got[lin], want[lin])


def test_golden_is_the_reference_curve_known_answers():
    """sanity of the fixture itself: closed-form values the sRGB standard fixes"""
    x, y = G['x'], G['linear2srgb']
    assert y[x <= 0].max() == 0.0 and abs(float(y[x >= 1].min()) - 1.0) < 1e-7 and np.all(np.diff(y[np.argsort(x)]) >= -1e-6)
    i = int(np.argmin(np.abs(x - np.float32(0.0031308))))
    assert x[i] == np.float32(0.0031308) and y[i] == np.float32(0.0031308) * np.float32(12.92)


def test_oracle_restatement_matches_reference_outputs():
    x = torch.from_numpy(G['x'])
    got = od.linear2srgb(x).numpy()
    assert _l2s_close(got, G['linear2srgb'])
    _linear_branch_exact(G['x'], got, 0.0031308, True)
    with np.errstate(invalid='ignore'):
        got = od.srgb2linear(torch.from_numpy(G['x_with_nan'])).numpy()
    assert _ulp_diff(got, G['srgb2linear_with_nan']).max() <= S2L_ULPS
    _linear_branch_exact(G['x'], got[:-1], 0.04045, False)
    # fp64 callers
    np.testing.assert_allclose(od.linear2srgb(torch.from_numpy(G['x'].astype(np.float64))).numpy(), G['linear2srgb_f64'], rtol=1e-14, atol=0)
    ok = ~np.isnan(G['srgb2linear_f64'])
    np.testing.assert_allclose(od.srgb2linear(torch.from_numpy(G['x'].astype(np.float64))).numpy()[ok], G['srgb2linear_f64'][ok], rtol=1e-14, atol=0)


def test_product_torch_statement_matches_reference_outputs():
    x = torch.from_numpy(G['x'])
    got = imgutil.linear2srgb(x).numpy()
    assert _l2s_close(got, G['linear2srgb'])
    _linear_branch_exact(G['x'], got, 0.0031308, True)
    got = imgutil.srgb2linear(torch.from_numpy(G['x_with_nan'])).numpy()
    assert _ulp_diff(got, G['srgb2linear_with_nan']).max() <= S2L_ULPS
    _linear_branch_exact(G['x'], got[:-1], 0.04045, False)


def test_output_path_helpers_match_reference_outputs():
    np.testing.assert_array_equal(visutil.to_uint8(G['x']), G['to_uint8'])                     # truncating cast of 255 * clip(x)
    np.testing.assert_array_equal(np.clip(G['x'], 0.0, 1.0), G['clip_0to1'])
    a, b, al = G['blend_a'], G['blend_b'], G['blend_alpha']
    np.testing.assert_array_equal(visutil.alpha_blend(a, al, b), G['alpha_blend_two'])
    np.testing.assert_array_equal(visutil.alpha_blend(a, al, np.zeros_like(a)), G['alpha_blend_zero_bg'])


@pytest.mark.gpu
def test_fused_kernel_matches_reference_outputs():
    from vqnerf_release_amd import _C
    dev = torch.device('cuda:0')
    x = torch.from_numpy(G['x_with_nan']).to(dev)
    got = _C.linear2srgb(x).cpu().numpy()
    assert np.isnan(got[-1])                                                                   # the clip of the statement keeps a NaN
    assert _l2s_close(got[:-1], G['linear2srgb'])
    _linear_branch_exact(G['x'], got[:-1], 0.0031308, True)
    assert torch.equal(imgutil.linear2srgb(x[:-1]).cpu(), torch.from_numpy(got[:-1]))          # the inference path IS the kernel
