"""torch.optim.Adam with the update as ONE HIP launch (csrc/adam.hip: vqn_adam_step).

Same constructor, same state (exp_avg, exp_avg_sq, max_exp_avg_sq, step as device tensors -- `capturable=True` is implied), same
state_dict, so checkpoints of the reference-shaped trainers load either way; `step()` is what differs: torch's fused multi-tensor
kernel gives each workgroup a 65,536-element chunk (16 workgroups and 85 us per launch for the ~1 M reflectance parameters, two
launches), this one 1,024.  Parameters that are not contiguous f32 device tensors fall back to torch's own update."""
import ctypes

import numpy as np
import torch

from vqnerf_release_amd import _C


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize,
                         capturable=True, fused=True)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps, gs, ms, vs, vmaxs, steps = [], [], [], [], [], []
            self._init_group(group, ps, gs, ms, vs, vmaxs, steps)
            if not ps:
                continue
            ok = all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous() and g.dtype == torch.float32 and
                     s.is_cuda and s.dtype == torch.float32 for p, g, s in zip(ps, gs, steps))
            if not ok:
                return super().step(closure=None) if loss is None else loss
            torch._foreach_add_(steps, 1)
            k = len(ps)
            arr = lambda ts: (ctypes.c_void_p * k)(*[t.data_ptr() for t in ts])
            n = np.array([p.numel() for p in ps], np.int64)
            lr = group['lr']
            b1, b2 = group['betas']
            with _C._clock('vqn_adam_step'):
                rc = _C.lib().vqn_adam_step(ctypes.c_int(k), arr(ps), arr(gs), arr(ms), arr(vs), arr(vmaxs) if group['amsgrad'] else None,
                                            arr(steps), n.ctypes.data_as(ctypes.c_void_p),
                                            _C._ptr(lr) if torch.is_tensor(lr) else None,
                                            ctypes.c_double(0.0 if torch.is_tensor(lr) else float(lr)), ctypes.c_double(b1), ctypes.c_double(b2),
                                            ctypes.c_double(group['eps']), ctypes.c_double(group['weight_decay']),
                                            ctypes.c_int(int(group['maximize'])), _C._stream())
            _C._check(rc, 'vqn_adam_step')
        import vqnerf_release_amd
        vqnerf_release_amd.weights_changed()           # (the global optimiser hook covers step(); kept explicit for direct callers)
        return loss
