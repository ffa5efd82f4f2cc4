"""Light-probe geometry: mirror of `gen_light_xyz` (decomp/nerfvq_nfr3/brdf/renderer.py:184-219; identical twin at
geo/NeuS-ours2/models/util.py:84-119).  Pure numpy, runs once per model; pinned by tests/golden/light_xyz_16x32.npz."""
import numpy as np


def gen_light_xyz(envmap_h, envmap_w, envmap_radius=1e2):
    """Centres and solid angles of an equirectangular `envmap_h x envmap_w` grid that excludes the poles.
    Returns (xyz [h,w,3], areas [h,w]); areas sum to 4*pi."""
    lat_step = np.pi / (envmap_h + 2)
    lng_step = 2 * np.pi / (envmap_w + 2)
    lats = np.linspace(np.pi / 2 - lat_step, -np.pi / 2 + lat_step, envmap_h)
    lngs = np.linspace(np.pi - lng_step, -np.pi + lng_step, envmap_w)
    lngs, lats = np.meshgrid(lngs, lats)
    r = envmap_radius
    xyz = np.dstack((r * np.cos(lats) * np.cos(lngs), r * np.cos(lats) * np.sin(lngs), r * np.sin(lats)))
    sin_colat = np.sin(np.pi / 2 - lats)
    areas = 4 * np.pi * sin_colat / np.sum(sin_colat)
    return xyz, areas
