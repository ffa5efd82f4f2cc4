"""vqnerf_release_amd -- MI355X-native (gfx950) implementation of the VQ-NeRF hot path.

Layout (only what the hot path needs):
    csrc/        hand-written HIP kernels + the C ABI declared in include/vqnerf_hip.h
    lib/         the built libvqnerf_hip.so (git-ignored, travels to the GPU box)
    _C.py        ctypes binding of that C ABI (fails loudly when the library is missing)
    geo/         host-side mirror of geo/NeuS-ours2/models/{renderer,fields,embedder}.py
    decomp/      host-side mirror of decomp/nerfvq_nfr3/nerfactor/{networks,models,util}
"""
__version__ = '0.1.0'


# ---- weights epoch -------------------------------------------------------------------------------------------------------
# The inference paths keep the networks' parameters re-laid as MFMA fragments ("packs") and rebuild a pack when a parameter
# changes.  torch's per-tensor `_version` counter does NOT see every write: the fused / capturable Adam
# (`torch._fused_adam_`) leaves it untouched, so does a replayed HIP graph, so does any raw-pointer writer behind the C ABI.
# Every pack cache therefore also keys on this process-wide counter.  It advances after every `optimizer.step()` of any torch
# optimiser (global post-step hook below), after every graph replay of the package's trainers, and on request.
_weights_epoch = [0]


def weights_epoch():
    return _weights_epoch[0]


def weights_changed():
    """Invalidate every cached weight pack / codebook-fragment image of the process.  Call it after rewriting parameters in a
    way torch cannot see (replaying a captured graph of your own, a ctypes kernel or DLPack peer writing into a parameter)."""
    _weights_epoch[0] += 1


def _install_optimizer_hook():
    try:
        from torch.optim.optimizer import register_optimizer_step_post_hook
        register_optimizer_step_post_hook(lambda *_a, **_k: weights_changed())
    except Exception:                     # noqa: BLE001  (an older torch: trainers of this package still call weights_changed())
        pass


_install_optimizer_hook()
