"""Dataset classes by name (mirror of decomp/nerfvq_nfr3/nerfactor/datasets/__init__.py:17-20)."""
from importlib import import_module


def get_dataset_class(name):
    return import_module('vqnerf_release_amd.decomp.nerfactor.datasets.' + name).Dataset
