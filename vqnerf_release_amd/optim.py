"""The reference's two optimisers with the update as ONE HIP launch (csrc/adam.hip: vqn_adam_step).

`HipAdam(params, ..., eps_mode='torch')` is `torch.optim.Adam` (geo/NeuS-ours2/nerf_runner.py:72); `eps_mode='keras'` is
`tf.keras.optimizers.Adam(amsgrad=True)` of decomp/nerfvq_nfr3/nerfactor/train_nfr.py:127-138 -- the same moments, but epsilon is
added to the UN-debiased sqrt(vhat): p -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(vhat) + eps) (TF 2.4.1
ResourceApplyAdamWithAmsgrad; oracle/optim.py states both).  Same constructor, same state (exp_avg, exp_avg_sq, max_exp_avg_sq, step), same
state_dict as torch.optim.Adam, so checkpoints of the reference-shaped trainers load either way: `load_state_dict` puts the step
counters and the learning rate back on the device after a state written by an eager (host-counter) Adam has been loaded.

`step()` is what differs: torch's fused multi-tensor kernel gives each workgroup a 65,536-element chunk (16 workgroups and 85 us per
launch for the ~1 M reflectance parameters, two launches), this one 1,024.  Eligibility is decided for ALL groups before any is touched;
when some tensor is not a contiguous f32 device tensor (CPU parameters in the gloo tests, a host-side lr) the whole step takes the
framework statement of the same update -- torch's own for eps_mode 'torch', `_keras_statement` for 'keras'."""
import ctypes
import math

import numpy as np
import torch

from vqnerf_release_amd import _C

EPS_MODES = {'torch': 0, 'keras': 1}


class HipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, maximize=False, eps_mode='torch',
                 capturable=None):
        assert eps_mode in EPS_MODES, eps_mode
        params = list(params)
        flat = [p for g in params for p in g['params']] if params and isinstance(params[0], dict) else params
        on_dev = bool(flat) and all(p.is_cuda for p in flat)
        if capturable is None:
            capturable = on_dev                  # step counters (and a tensor lr) on the device: what a captured training step needs
        self.eps_mode = eps_mode
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, maximize=maximize,
                         capturable=bool(capturable), fused=True if (capturable and on_dev) else None)
        self._ctor_flags = (bool(capturable), True if (capturable and on_dev) else None)
        # the epsilon placement travels with the checkpoint (param_groups are part of state_dict): an optimiser built with the other mode
        # must not silently train on a different update (31.6x effective epsilon at step 1)
        self.defaults['eps_mode'] = eps_mode
        for g in self.param_groups:
            g['eps_mode'] = eps_mode

    # ------------------------------------------------------------------ checkpoints
    def load_state_dict(self, state_dict):
        """torch replaces the groups' hyper-parameters with the saved ones: a checkpoint of an eager Adam brings a python-float lr,
        capturable=False and host step counters.  Keep THIS optimiser's execution mode: lr value copied into the existing device
        scalar, capturable / fused as constructed, every state['step'] a float32 tensor on its parameter's device."""
        lr_before = [g['lr'] for g in self.param_groups]
        saved_modes = {g.get('eps_mode', self.eps_mode) for g in state_dict.get('param_groups', [])}     # (absent: a plain torch.optim.Adam state)
        if saved_modes - {self.eps_mode}:
            raise ValueError(f"checkpoint was written by an Adam with eps_mode={sorted(saved_modes)}, this optimiser was built with "
                             f"eps_mode='{self.eps_mode}': build it with the same mode (torch: eps inside the debiased root; keras: outside)")
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            g['eps_mode'] = self.eps_mode
        capturable, fused = self._ctor_flags
        for g, lr0 in zip(self.param_groups, lr_before):
            g['capturable'], g['fused'] = capturable, fused
            if torch.is_tensor(lr0):
                new = g['lr']
                with torch.no_grad():
                    lr0.copy_(new.to(lr0.device) if torch.is_tensor(new) else torch.tensor(float(new)))
                g['lr'] = lr0
            elif torch.is_tensor(g['lr']):
                g['lr'] = float(g['lr'])
            for p in g['params']:
                st = self.state.get(p)
                if st and 'step' in st:
                    s = st['step']
                    dev = p.device if capturable else torch.device('cpu')
                    st['step'] = (s.detach().to(device=dev, dtype=torch.float32) if torch.is_tensor(s)
                                  else torch.tensor(float(s), dtype=torch.float32, device=dev))

    # ------------------------------------------------------------------ the step
    @staticmethod
    def _eligible(group, ps, gs, steps):
        lr = group['lr']
        if torch.is_tensor(lr) and not (lr.is_cuda and lr.dtype == torch.float32 and lr.numel() == 1):
            return False
        return all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous() and g.dtype == torch.float32
                   and not g.is_sparse and s.is_cuda and s.dtype == torch.float32 for p, g, s in zip(ps, gs, steps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        work = []
        for group in self.param_groups:
            ps, gs, ms, vs, vmaxs, steps = [], [], [], [], [], []
            self._init_group(group, ps, gs, ms, vs, vmaxs, steps)
            work.append((group, ps, gs, ms, vs, vmaxs, steps))
        if not all(self._eligible(w[0], w[1], w[2], w[6]) for w in work if w[1]):
            # nothing has been touched yet: ONE framework-side update of every group (the closure has already been evaluated)
            if self.eps_mode == 'keras':
                for w in work:
                    self._keras_statement(*w)
            else:
                super().step(closure=None)
            return loss
        for group, ps, gs, ms, vs, vmaxs, steps in work:
            if not ps:
                continue
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
                                            ctypes.c_int(int(group['maximize'])), ctypes.c_int(EPS_MODES[self.eps_mode]), _C._stream())
            _C._check(rc, 'vqn_adam_step')
        import vqnerf_release_amd
        vqnerf_release_amd.weights_changed()           # (the global optimiser hook covers step(); kept explicit for direct callers)
        return loss

    @staticmethod
    def _keras_statement(group, ps, gs, ms, vs, vmaxs, steps):
        """Keras Adam on framework ops (CPU parameters: the gloo tests; oracle/optim.py: keras_adam_step is the numpy statement)."""
        b1, b2 = group['betas']
        eps, lr = group['eps'], group['lr']
        for i, (p, g, m, v, s) in enumerate(zip(ps, gs, ms, vs, steps)):
            s += 1
            t = float(s)
            if group['maximize']:
                g = -g
            if group['weight_decay'] != 0.0:
                g = g + group['weight_decay'] * p
            alpha = (float(lr) if not torch.is_tensor(lr) else lr) * (math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
            m.add_((g - m) * (1.0 - b1))
            v.add_((g * g - v) * (1.0 - b2))
            if group['amsgrad']:
                torch.maximum(vmaxs[i], v, out=vmaxs[i])
                den = vmaxs[i].sqrt() + eps
            else:
                den = v.sqrt() + eps
            p.sub_(m * alpha / den)
