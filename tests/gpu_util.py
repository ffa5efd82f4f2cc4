"""Helpers shared by the GPU tests."""


class launches:
    """`with launches() as rec: ...` -- names of the C-ABI entry points launched inside the block (the package's
    `_C.KernelClock`, HIP events on the launch stream).  GPU tests use it to prove that a 'hip' backend really ran the HIP
    kernels it claims (a comparison that silently fell back to torch ops would compare torch with torch)."""

    def __enter__(self):
        from vqnerf_release_amd import _C
        _C.KernelClock.reset(True)
        self.names = set()
        return self

    def __exit__(self, *exc):
        import torch
        from vqnerf_release_amd import _C
        torch.cuda.synchronize()
        self.names = set(_C.KernelClock.summary())
        self.counts = {k: v[0] for k, v in _C.KernelClock.summary().items()}
        _C.KernelClock.reset(False)
        return False

    def ran(self, prefix):
        return any(n.startswith(prefix) for n in self.names)
