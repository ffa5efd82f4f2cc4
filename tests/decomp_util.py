"""Shared helpers for the decomp tests: a vq_nfr.ini-shaped config and oracle-parameter loading."""
import numpy as np
import torch

from vqnerf_release_amd.decomp.nerfactor.util.io import config_from_dict

# the keys vq_nfr.Model reads, with the shipped values (decomp/nerfvq_nfr3/nerfactor/config/vq_nfr.ini)
VQ_NFR_INI = dict(
    model='vq_nfr', dataset='shape_unit', data_type='nerf', white_bg='True', pred_brdf='True', mlp_chunk=100000, mlp_width=128,
    mlp_depth=4, mlp_skip_at=2, conv_width=256, pos_enc='True', n_freqs_xyz=10, n_freqs_ldir=4, n_freqs_vdir=4,
    light_h=16, light_init_val=0.5, num_embed=15, commitment_cost=0.1, vq_loss_weight=1.0, chr_alpha=60, chr_thres=0.1,
    combine_weight=0.2, mat_sloss_weight=0.05, chromaticity_loss_weight=1.0, sim_loss_weight=1e-4, lambert_weight=1e-3,
    no_brdf_chunk='True', random_seed=2, l_var_weight=0.0, n_rays_per_step=1024, lr=5e-4,
    cluster_center_path='', nfr_model_ckpt='', test_envmap_dir='',
)


def make_config(**over):
    d = dict(VQ_NFR_INI)
    d.update(over)
    return config_from_dict(d)


def load_oracle_params(model, p, device):
    """oracle.decomp.make_model_params -> vq_nfr.Model (Keras layout on both sides)."""
    model.build_nets(device=device, seed=0)
    with torch.no_grad():
        for name, net in model.net.items():
            for layer, (W, b) in zip(net.layers, p[name]):
                layer.kernel.copy_(torch.as_tensor(np.asarray(W)))
                layer.bias.copy_(torch.as_tensor(np.asarray(b)))
    model.to(device)
    if 'codebook_raw' in p and hasattr(model, 'set_codebook'):
        model.set_codebook(np.asarray(p['codebook_raw']).T)
    model.set_light(np.asarray(p['light']))
    return model


def make_batch(pts, device, bg_every=0):
    """oracle.decomp.make_points dict -> the batch tuple of datasets/shape_unit.py:109-110 (data_type nerf)."""
    n = pts['xyz'].shape[0]
    T = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=device)
    alpha = torch.ones(n, 1, device=device)
    if bg_every:
        alpha[::bg_every] = 0.0
    id_ = ['view'] * n
    hw = torch.zeros(n, 2, device=device)
    rayd = torch.zeros(n, 3, device=device)
    batch = (id_, hw, T(pts['rayo']), rayd, T(pts['rgb']), alpha, alpha.clone(), T(pts['xyz']), T(pts['normal']))
    if 'lvis' in pts:
        batch = batch + (T(pts['lvis']),)
    return batch
