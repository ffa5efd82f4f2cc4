"""Model registry: mirror of decomp/nerfvq_nfr3/nerfactor/models/__init__.py (`get_model_class(name).Model`)."""
from importlib import import_module


def get_model_class(name):
    mod = import_module('vqnerf_release_amd.decomp.nerfactor.models.' + name)
    return mod.Model
