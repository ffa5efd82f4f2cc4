"""ORACLE (test infrastructure, never imported by the product path): the optimiser update statements of the reference's two trainers,
in plain numpy float32, one tensor at a time.

* `torch_adam_step` -- `torch.optim.Adam(params, lr)` as constructed at geo/NeuS-ours2/nerf_runner.py:72 (defaults: betas (0.9, 0.999),
  eps 1e-8, no weight decay, no AMSGrad).  The statement is torch's `_single_tensor_adam` (torch/optim/adam.py):
      m <- m + (1 - b1)(g - m);  v <- b2 v + (1 - b2) g g;  p <- p - (lr / (1 - b1^t)) m / (sqrt(vhat) / sqrt(1 - b2^t) + eps)
* `keras_adam_step` -- `tf.keras.optimizers.Adam(learning_rate, amsgrad=True[, clipnorm | clipvalue])` as constructed at
  decomp/nerfvq_nfr3/nerfactor/train_nfr.py:127-138 (Keras defaults: beta_1 0.9, beta_2 0.999, epsilon 1e-7).  THIRD-PARTY arithmetic:
  TensorFlow 2.4.1 is not installable in this image, so this restates its published kernel -- `ApplyAdamWithAmsgrad` of
  tensorflow/core/kernels/training_ops.cc, reached from keras/optimizer_v2/adam.py `_resource_apply_dense`:
      alpha = lr sqrt(1 - b2^t) / (1 - b1^t);  m <- m + (g - m)(1 - b1);  v <- v + (g g - v)(1 - b2);  vhat <- max(vhat, v);
      p <- p - (m alpha) / (sqrt(vhat) + eps)
  The difference that matters: Keras adds eps to the UN-debiased sqrt(vhat), torch to the debiased one -- an effective epsilon
  larger by 1 / sqrt(1 - b2^t) (31.6 x at t = 1, 1.0 for t >> 1000).  It only shows for |g| <~ eps.
  Parity of this statement with TensorFlow itself is UNPINNED (no TF here, the reference ships no optimiser fixtures); what pins it is
  the closed forms in tests/test_oracle_optim.py (first-step size lr g / (|g| + eps sqrt(1 - b2)) etc.).
"""
import numpy as np

f32 = np.float32


def torch_adam_step(p, g, m, v, vmax, t, lr, b1=0.9, b2=0.999, eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False):
    """One update; t = step count AFTER the increment (1 at the first step).  Returns (p, m, v, vmax) as float32 arrays."""
    p, g, m, v = (np.asarray(a, f32) for a in (p, g, m, v))
    if maximize:
        g = -g
    if weight_decay != 0.0:
        g = (g + f32(weight_decay) * p).astype(f32)
    m = (m + f32(1.0 - b1) * (g - m)).astype(f32)                       # exp_avg.lerp_(grad, 1 - beta1)
    v = (f32(b2) * v + f32(1.0 - b2) * g * g).astype(f32)               # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1.0 - b1 ** t
    bc2_sqrt = np.sqrt(1.0 - b2 ** t)
    step_size = f32(lr / bc1)
    if amsgrad:
        vmax = np.maximum(np.asarray(vmax, f32), v)
        denom = (np.sqrt(vmax) / f32(bc2_sqrt) + f32(eps)).astype(f32)
    else:
        denom = (np.sqrt(v) / f32(bc2_sqrt) + f32(eps)).astype(f32)
    p = (p - step_size * (m / denom)).astype(f32)
    return p, m, v, vmax


def keras_adam_step(p, g, m, v, vhat, t, lr, b1=0.9, b2=0.999, eps=1e-7, amsgrad=True):
    """One update of tf.keras.optimizers.Adam (TF 2.4.1 kernels); t = `iterations + 1`.  Returns (p, m, v, vhat)."""
    p, g, m, v = (np.asarray(a, f32) for a in (p, g, m, v))
    alpha = f32(lr) * f32(np.sqrt(f32(1.0) - f32(b2) ** f32(t))) / (f32(1.0) - f32(b1) ** f32(t))
    m = (m + (g - m) * f32(1.0 - b1)).astype(f32)
    v = (v + (g * g - v) * f32(1.0 - b2)).astype(f32)
    if amsgrad:
        vhat = np.maximum(np.asarray(vhat, f32), v)
        p = (p - (m * f32(alpha)) / (np.sqrt(vhat) + f32(eps))).astype(f32)
    else:
        p = (p - (m * f32(alpha)) / (np.sqrt(v) + f32(eps))).astype(f32)
    return p, m, v, vhat


def clip_grads_keras(grads, clipnorm=-1.0, clipvalue=-1.0):
    """Keras `clipnorm` clips EACH gradient tensor to that l2 norm (tf.clip_by_norm per tensor, optimizer_v2.py `_clip_gradients` of TF 2.4),
    `clipvalue` clamps element-wise."""
    out = []
    for g in grads:
        g = np.asarray(g, f32)
        if clipnorm > 0:
            n = np.sqrt(np.sum(g.astype(np.float64) ** 2))
            g = (g * f32(clipnorm / max(n, clipnorm))).astype(f32)
        if clipvalue > 0:
            g = np.clip(g, -clipvalue, clipvalue)
        out.append(g)
    return out
