"""Stage-3 loader: mirror of decomp/nerfvq_nfr3/nerfactor/datasets/ref_nfr.py -- the shape_unit view with one more column, `ref`: the
reference colour of every pixel, read from `<data_nerf_root>/<id>/rgb.png` (ref_nfr.py:64-68 `basecolor_path`, :258-259 load +
normalize_uint, :272-273 resize) and handed to the model between `normal` and `lvis` (ref_nfr.py:299-301; the model's batch layout
`id, hw, rayo, rayd, rgb, alpha, pred_alpha, xyz, normal, ref[, lvis]`, models/ref_nfr.py:180-184).  Everything else -- file discovery,
ray generation, compositing, the device-resident cache -- is datasets/shape_unit.py; the pair sampling of the reference's
`_sample_rays` (:107-183, the max-colour-difference neighbour) is train_nfr.outer_sample(neighbour='max_diff'), which gathers every tensor
column of the view, `ref` included."""
from os.path import join

import numpy as np

from vqnerf_release_amd.decomp.nerfactor.datasets import shape_unit


class Dataset(shape_unit.Dataset):
    def _extra_paths(self, nerf_root, id_):
        return {'basecolor': join(nerf_root, id_, 'rgb.png')}

    def _extra_maps(self, paths, fit):
        ref = shape_unit.read_image_normalized(paths['basecolor'])
        if ref.ndim == 2:
            ref = np.repeat(ref[:, :, None], 3, 2)
        return (fit(ref[:, :, :3]).astype(np.float32),)
