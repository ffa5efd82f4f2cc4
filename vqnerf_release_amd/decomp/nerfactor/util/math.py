"""Mirror of the pieces of decomp/nerfvq_nfr3/nerfactor/util/math.py the hot path uses (torch statement)."""
import torch


def safe_l2_normalize(x, axis=None, eps=1e-6):
    """tf.linalg.l2_normalize (math.py:63-64): x * rsqrt(max(sum(x^2, axis), eps))."""
    sq = (x * x).sum(dim=axis, keepdim=True) if axis is not None else (x * x).sum()
    return x * torch.rsqrt(torch.clamp(sq, min=eps))


class _ClipPreserve(torch.autograd.Function):
    """One launch (vqn_clip_preserve: the same three roundings) instead of clamp / sub / add; the gradient passes through unchanged."""

    @staticmethod
    def forward(ctx, x, lo, hi):
        from vqnerf_release_amd import _C
        return _C.clip_preserve(x.detach().contiguous(), lo, hi)

    @staticmethod
    def backward(ctx, g):
        return g, None, None


def clip_preserve_gradient(x, lo, hi):
    """tfp.math.clip_by_value_preserve_gradient: clipped value, identity gradient -- in the reference's own arithmetic,
    `x + stop_gradient(clip(x) - x)` (which is not always bitwise `clip(x)`)."""
    if x.is_cuda and x.dtype == torch.float32:
        return _ClipPreserve.apply(x, float(lo), float(hi))
    return x + (x.clamp(lo, hi) - x).detach()


def divide_no_nan(a, b):
    """tf.math.divide_no_nan: 0 where b == 0 (value and gradient)."""
    zero = b == 0
    return torch.where(zero, torch.zeros_like(a * b), a / torch.where(zero, torch.ones_like(b), b))


class InvalidArgumentError(ValueError):
    """What `tf.debugging.check_numerics` raises in the reference (tf.errors.InvalidArgumentError)."""


def check_numerics(x, message):
    """tf.debugging.check_numerics: raise if `x` holds a NaN or an Inf, else return it.  One host sync per call, which is why the
    models only call it in debug mode (`Model(config, debug=True)` or VQN_CHECK_NUMERICS=1)."""
    if x is not None and torch.is_tensor(x) and x.numel() and not bool(torch.isfinite(x).all()):
        bad = 'NaN' if bool(torch.isnan(x).any()) else 'Inf'
        raise InvalidArgumentError(f'{message} : Tensor had {bad} values')
    return x
