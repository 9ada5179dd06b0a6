"""Mirror of reference models/pos_encoding.py:6-96 (``get_encoder``) on the gfx950 kernels.

Same signature and return value ``(embed, out_dim)``.  The reference's live configurations are
``'HashGrid'`` and ``'OneBlob'`` (configs/slam.yaml:19-26); the dense / spherical / frequency / identity
branches of the reference are never selected by any shipped config and raise here.
"""
import numpy as np
import torch

from . import tcnn_shim as tcnn


def get_encoder(encoding, input_dim=3, degree=4, n_bins=16, n_frequencies=12, n_levels=16, level_dim=2,
                base_resolution=16, log2_hashmap_size=19, desired_resolution=512):
    name = encoding.lower()
    if "hash" in name or "tiled" in name:
        # reference pos_encoding.py:33: float64 numpy
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (n_levels - 1))
        embed = tcnn.Encoding(
            n_input_dims=input_dim,
            encoding_config={"otype": "HashGrid", "n_levels": n_levels, "n_features_per_level": level_dim,
                             "log2_hashmap_size": log2_hashmap_size, "base_resolution": base_resolution,
                             "per_level_scale": per_level_scale},
            dtype=torch.float)
    elif "blob" in name:
        embed = tcnn.Encoding(n_input_dims=input_dim, encoding_config={"otype": "OneBlob", "n_bins": n_bins},
                              dtype=torch.float)
    else:
        raise ValueError(f"dns_slam_amd.get_encoder: encoding {encoding!r} is not on the supported hot path "
                         "(HashGrid, OneBlob)")
    return embed, embed.n_output_dims
