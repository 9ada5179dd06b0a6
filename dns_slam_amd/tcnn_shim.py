"""A ``tinycudann``-shaped module backed by the gfx950 kernels.

The reference reaches its only native code through ``import tinycudann as tcnn`` and two constructors
(reference models/pos_encoding.py:16-94, models/decoder.py:58,84,101,110, slams/mapping.py:737):

    tcnn.Encoding(n_input_dims, encoding_config, dtype=torch.float)   -> module with .n_output_dims
    tcnn.Network(n_input_dims, n_output_dims, network_config)         -> module, flat fp32 ``.params``

Registering this module as ``sys.modules['tinycudann']`` before importing the reference makes
``models/decoder.py`` and ``slams/mapping.py`` run unchanged on MI355X.  Supported
configs are the ones the reference uses: Encoding otype HashGrid / OneBlob; Network otype CutlassMLP or
FullyFusedMLP, ReLU, no output activation, 32 or 64 neurons, 1 or 2 hidden layers.  Networks compute and
return fp32 (tcnn returns fp16; the reference immediately calls ``.float()``, models/decoder.py:94).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import ops


class Encoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, dtype=torch.float, seed=1337):
        super().__init__()
        if n_input_dims != 3:
            raise ValueError("dns_slam_amd.tcnn_shim.Encoding: only 3-D inputs are supported")
        otype = str(encoding_config.get("otype", "")).lower()
        self.n_input_dims = n_input_dims
        self.encoding_config = dict(encoding_config)
        self.dtype = dtype
        self.seed = seed
        if otype == "oneblob":
            self.kind = "oneblob"
            self.n_bins = int(encoding_config.get("n_bins", 16))
            self.n_output_dims = n_input_dims * self.n_bins
            self.meta = None
            self.params = nn.Parameter(torch.zeros(0), requires_grad=False)
        elif otype in ("hashgrid", "grid"):
            if str(encoding_config.get("type", "Hash")).lower() != "hash" and otype == "grid":
                raise ValueError("dns_slam_amd.tcnn_shim.Encoding: only hashed grids are supported")
            self.kind = "hashgrid"
            self.n_bins = 0
            L = int(encoding_config.get("n_levels", 16))
            F = int(encoding_config.get("n_features_per_level", 2))
            self.meta = ops.GridMeta(int(encoding_config.get("log2_hashmap_size", 19)), 0, L, F,
                                     int(encoding_config.get("base_resolution", 16)),
                                     per_level_scale=float(encoding_config.get("per_level_scale", 2.0)))
            self.n_output_dims = L * F
            g = torch.Generator().manual_seed(seed)
            # tcnn grid initialisation: U(-1e-4, 1e-4)
            init = (torch.rand(self.meta.total_rows * F, generator=g) * 2 - 1) * 1e-4
            self.params = nn.Parameter(init.float())
        else:
            raise ValueError(f"dns_slam_amd.tcnn_shim.Encoding: unsupported otype {encoding_config.get('otype')!r}")

    def forward(self, x):
        x = x.float()
        if self.kind == "oneblob":
            return ops.encode(x, None, None, None, self.n_bins, True, False)
        return ops.encode(x, self.params, self.meta, None, 16, False, True)


class Network(nn.Module):
    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        super().__init__()
        act = str(network_config.get("activation", "ReLU")).lower()
        oact = str(network_config.get("output_activation", "None")).lower()
        if act != "relu" or oact != "none":
            raise ValueError("dns_slam_amd.tcnn_shim.Network: only ReLU hidden / no output activation is supported")
        self.n_input_dims = int(n_input_dims)
        self.n_output_dims = int(n_output_dims)
        self.n_neurons = int(network_config.get("n_neurons", 32))
        self.n_hidden_layers = int(network_config.get("n_hidden_layers", 1))
        self.network_config = dict(network_config)
        self.fp16 = str(network_config.get("dtype", "fp32")).lower() in ("fp16", "half", "float16")
        self.seed = seed
        n = ops.mlp_param_count(self.n_input_dims, self.n_output_dims, self.n_neurons, self.n_hidden_layers)
        self.params = nn.Parameter(self._xavier(seed))
        assert self.params.numel() == n

    def _shapes(self):
        nn_, i, o = self.n_neurons, self.n_input_dims, ops.mlp_out_padded(self.n_output_dims)
        return [(nn_, i)] + [(nn_, nn_)] * (self.n_hidden_layers - 1) + [(o, nn_)]

    def _xavier(self, seed):
        g = torch.Generator().manual_seed(seed)
        chunks = []
        for (r, c) in self._shapes():
            s = math.sqrt(6.0 / (r + c))
            chunks.append((torch.rand(r * c, generator=g) * 2 - 1) * s)
        return torch.cat(chunks).float()

    def forward(self, x):
        return ops.mlp(x, self.params, self.n_input_dims, self.n_output_dims, self.n_neurons, self.n_hidden_layers,
                       fp16=getattr(self, "fp16", False))
