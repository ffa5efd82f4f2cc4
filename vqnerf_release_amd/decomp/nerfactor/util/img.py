"""sRGB transfer functions: mirror of decomp/nerfvq_nfr3/nerfactor/util/img.py:142-186 (torch statement)."""
import torch

_THRES_L2S, _THRES_S2L, _LIN, _EXP_C, _EXP = 0.0031308, 0.04045, 12.92, 1.055, 2.4


def linear2srgb(t):
    if t.is_cuda and t.dtype == torch.float32 and not (torch.is_grad_enabled() and t.requires_grad):
        from vqnerf_release_amd import _C                   # the inference path: one fused pass (vqn_linear2srgb)
        return _C.linear2srgb(t)
    t = t.clamp(0.0, 1.0)                                   # img.py:155 (_clip_0to1_warn)
    return torch.where(t <= _THRES_L2S, t * _LIN, _EXP_C * torch.pow(t, 1.0 / _EXP) - (_EXP_C - 1.0))


def srgb2linear(t):
    return torch.where(t <= _THRES_S2L, t / _LIN, torch.pow((t + _EXP_C - 1.0) / _EXP_C, _EXP))
