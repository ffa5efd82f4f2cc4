"""vqnerf_release_amd -- MI355X-native (gfx950) implementation of the VQ-NeRF hot path.

Layout (only what the hot path needs):
    csrc/        hand-written HIP kernels + the C ABI declared in include/vqnerf_hip.h
    lib/         the built libvqnerf_hip.so (git-ignored, travels to the GPU box)
    _C.py        ctypes binding of that C ABI (fails loudly when the library is missing)
    geo/         host-side mirror of geo/NeuS-ours2/models/{renderer,fields,embedder}.py
    decomp/      host-side mirror of decomp/nerfvq_nfr3/nerfactor/{networks,models,util}
"""
__version__ = '0.1.0'
