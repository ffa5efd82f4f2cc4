"""CPU: the oracle statements of the two optimisers (oracle/optim.py) against closed forms and against torch.optim.Adam; the
host-side behaviour of optim.HipAdam on CPU parameters (the framework statement it falls back to as a WHOLE, checkpoints of an eager
Adam loaded into a capturable one: ADVICE r03)."""
import numpy as np
import pytest
import torch

from oracle import optim as oo
from vqnerf_release_amd.optim import HipAdam


def test_keras_first_step_closed_form():
    """t = 1, m0 = v0 = 0: m = (1-b1) g, v = (1-b2) g^2, alpha = lr sqrt(1-b2)/(1-b1)  =>  dp = -lr g / (|g| + eps / sqrt(1-b2)).
    torch's placement gives -lr g / (|g| + eps) instead: the two differ visibly for |g| ~ eps."""
    lr, eps, b2 = 1e-3, 1e-7, 0.999
    g = np.array([1.0, -0.5, 1e-6, 1e-7, -3e-8, 0.0], np.float32)
    z = np.zeros_like(g)
    pk, _, _, _ = oo.keras_adam_step(z, g, z, z, z, 1, lr, eps=eps, amsgrad=True)
    pt, _, _, _ = oo.torch_adam_step(z, g, z, z, z, 1, lr, eps=eps, amsgrad=True)
    g64 = g.astype(np.float64)
    want_k = -lr * g64 / (np.abs(g64) + eps / np.sqrt(1 - b2))
    want_t = -lr * g64 / (np.abs(g64) + eps)
    assert np.allclose(pk, want_k, rtol=2e-5, atol=1e-12)
    assert np.allclose(pt, want_t, rtol=2e-5, atol=1e-12)
    assert abs(pk[3] / pt[3] - (1e-7 + 1e-7) / (1e-7 + 1e-7 / np.sqrt(1 - b2))) < 1e-3      # |g| = eps: 0.0613 of torch's step


def test_torch_statement_is_torch_adam():
    rng = np.random.default_rng(0)
    for amsgrad, wd in [(False, 0.0), (True, 0.0), (True, 0.01)]:
        p0 = rng.standard_normal(257).astype(np.float32)
        p = torch.nn.Parameter(torch.tensor(p0))
        opt = torch.optim.Adam([p], lr=3e-3, eps=1e-8, amsgrad=amsgrad, weight_decay=wd)
        q, m, v, vm = p0.copy(), np.zeros_like(p0), np.zeros_like(p0), np.zeros_like(p0)
        for t in range(1, 6):
            g = (rng.standard_normal(257) * 10.0 ** rng.integers(-8, 1)).astype(np.float32)
            p.grad = torch.tensor(g)
            opt.step()
            q, m, v, vm = oo.torch_adam_step(q, g, m, v, vm, t, 3e-3, eps=1e-8, weight_decay=wd, amsgrad=amsgrad)
            assert np.allclose(p.detach().numpy(), q, rtol=0, atol=2e-7 * max(1.0, np.abs(q).max())), (amsgrad, wd, t)


def test_keras_amsgrad_keeps_the_running_maximum_and_limits():
    """vhat never decreases; for t -> large and |g| >> eps both placements agree."""
    g_seq = [np.full(4, s, np.float32) for s in (1.0, 1e-3, 1e-3, 1e-3)]
    p = m = v = vh = np.zeros(4, np.float32)
    vh_prev = vh
    for t, g in enumerate(g_seq, 1):
        p, m, v, vh = oo.keras_adam_step(p, g, m, v, vh, t, 1e-2)
        assert (vh >= vh_prev).all()
        vh_prev = vh
    assert (v < vh).all()                                      # the later small gradients pulled v below its maximum
    pk, *_ = oo.keras_adam_step(np.zeros(3, np.float32), np.ones(3, np.float32), np.full(3, .5, np.float32), np.full(3, .25, np.float32),
                                np.full(3, .25, np.float32), 100000, 1e-3)
    pt, *_ = oo.torch_adam_step(np.zeros(3, np.float32), np.ones(3, np.float32), np.full(3, .5, np.float32), np.full(3, .25, np.float32),
                                np.full(3, .25, np.float32), 100000, 1e-3, eps=1e-7, amsgrad=True)
    assert np.allclose(pk, pt, rtol=1e-6)


def test_hip_adam_on_cpu_parameters_is_the_keras_statement_and_steps_every_group_once():
    """CPU parameters are not eligible for the kernel: the WHOLE step takes the framework statement (the old code could
    re-step earlier groups), with or without a closure (the old code returned the loss without updating), for both eps placements."""
    rng = np.random.default_rng(1)
    for mode in ('keras', 'torch'):
        a0, b0 = rng.standard_normal(33).astype(np.float32), rng.standard_normal((5, 7)).astype(np.float32)
        pa, pb = torch.nn.Parameter(torch.tensor(a0)), torch.nn.Parameter(torch.tensor(b0))
        opt = HipAdam([{'params': [pa]}, {'params': [pb], 'lr': 2e-3}], lr=1e-3, eps=1e-7, amsgrad=True, eps_mode=mode)
        assert not opt.param_groups[0]['capturable']
        st = {0: [a0.copy(), 0 * a0, 0 * a0, 0 * a0], 1: [b0.copy(), 0 * b0, 0 * b0, 0 * b0]}
        step = oo.keras_adam_step if mode == 'keras' else (lambda *a, **k: oo.torch_adam_step(*a, **k))
        for t in range(1, 4):
            ga = (rng.standard_normal(33) * (1e-7 if t == 2 else 1.0)).astype(np.float32)         # one step with |g| ~ eps
            gb = rng.standard_normal((5, 7)).astype(np.float32)
            calls = []

            def closure():
                calls.append(1)
                pa.grad, pb.grad = torch.tensor(ga), torch.tensor(gb)
                return torch.tensor(float(t))
            loss = opt.step(closure) if t != 3 else (closure(), opt.step())[0]
            assert len(calls) == 1 and float(loss) == float(t)
            for k, (g, lr) in enumerate(((ga, 1e-3), (gb, 2e-3))):
                p, m, v, vh = st[k]
                st[k] = list(step(p, g, m, v, vh, t, lr, eps=1e-7, amsgrad=True))
            for k, p in enumerate((pa, pb)):
                assert float(opt.state[p]['step']) == t                                        # once per call, every group
                assert np.allclose(p.detach().numpy(), st[k][0], rtol=0, atol=3e-7 * max(1.0, np.abs(st[k][0]).max())), (mode, t, k)


def test_an_eager_adam_checkpoint_loads_into_a_capturable_hip_adam():
    """torch's load_state_dict replaces lr / capturable / fused by the saved ones and leaves host step counters; the
    override keeps the optimiser's own execution mode (device-scalar lr with the loaded VALUE, tensor step counters on the parameter's
    device, capturable as constructed)."""
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(11))
    eager = torch.optim.Adam([w], lr=7e-4, eps=1e-7, amsgrad=True)
    for _ in range(3):
        w.grad = torch.randn(11)
        eager.step()
    sd = eager.state_dict()
    assert not torch.is_tensor(sd['param_groups'][0]['lr']) and sd['param_groups'][0]['capturable'] is False
    w2 = torch.nn.Parameter(w.detach().clone())
    lr_t = torch.tensor(1e-3)
    cap = HipAdam([w2], lr=lr_t, eps=1e-7, amsgrad=True, eps_mode='keras', capturable=True)
    cap.load_state_dict(sd)
    g = cap.param_groups[0]
    assert g['capturable'] is True and g['lr'] is lr_t and abs(float(lr_t) - 7e-4) < 1e-10
    s = cap.state[w2]['step']
    assert torch.is_tensor(s) and s.dtype == torch.float32 and s.device == w2.device and float(s) == 3.0
    assert torch.equal(cap.state[w2]['exp_avg'], eager.state[w]['exp_avg'])
    w2.grad = torch.randn(11)
    cap.step()                                                    # (CPU: the framework statement) -- runs, counts on
    assert float(cap.state[w2]['step']) == 4.0
    # and the other direction: a capturable state into an eager HipAdam keeps host-side execution
    w3 = torch.nn.Parameter(w.detach().clone())
    plain = HipAdam([w3], lr=1e-3, eps=1e-7, amsgrad=True, eps_mode='keras', capturable=False)
    plain.load_state_dict(cap.state_dict())
    assert plain.param_groups[0]['capturable'] is False and not torch.is_tensor(plain.param_groups[0]['lr'])
    assert float(plain.state[w3]['step']) == 4.0
    # the epsilon placement travels with the checkpoint: it is in param_groups, and the other mode refuses the state
    assert cap.state_dict()['param_groups'][0]['eps_mode'] == 'keras' and plain.param_groups[0]['eps_mode'] == 'keras'
    other = HipAdam([torch.nn.Parameter(w.detach().clone())], lr=1e-3, eps=1e-7, amsgrad=True, eps_mode='torch', capturable=False)
    with pytest.raises(ValueError, match="eps_mode"):
        other.load_state_dict(cap.state_dict())
    other.load_state_dict(sd)                                     # a plain torch.optim.Adam state carries no mode: accepted by either
