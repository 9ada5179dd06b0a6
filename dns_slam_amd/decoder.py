"""Mirror of reference models/decoder.py:7-125 (``Decoder`` and its parts) on the gfx950 kernels.

Same constructor, attribute names (``pe_fn``, ``coarse_fn``, ``out_fn``, ``merge``, ``pe_dim``, ``grid_dim``,
``pts_dim``, ``hidden_dim``, ``pixel_dim``, ``n_class``) and state-dict keys (``pe_fn.grid_fn.params``,
``coarse_fn.decoder.params``, ``out_fn.color_decoder.params``, ``out_fn.logit_decoder.params``,
``merge.decoder.params``) as the reference, so ``Mapper`` / ``Tracker`` / ``Mesher`` code written against the
reference class works against this one.  Differences, all inside the module bodies:
  * ``Pos_Encoding.forward`` is ONE kernel writing OneBlob and hash-grid channels into one [P, 80] buffer;
    the returned ``(pe, grid)`` are views of it, and ``Coarse`` re-uses the buffer instead of ``torch.cat``.
  * MLPs compute in fp32 on the matrix cores (tcnn: fp16) -- SURVEY D11.
  * extra cfg keys ``cfg['mlp'] = {'n_neurons': 32|64, 'n_hidden_layers': 1|2, 'dtype': 'fp32'|'fp16'}`` (default = the
    reference's 1x32; fp32 is the parity path; 'fp16' = tcnn's own precision, BASELINE configs[4]: through ``fused_step.MapStep``
    the networks then run on the HALF-ROWS kernels of ABI v12 -- f16 rows, f16 weight operands, fp32 accumulation, static loss
    scale ``mapper.loss_scale`` (128) --, through the autograd ops on the fp32-grade kernels with single f16 operands).
"""
import torch
from torch import nn

from . import ops
from . import tcnn_shim as tcnn
from .pos_encoding import get_encoder


def _mlp_cfg(cfg, hidden_dim):
    m = cfg.get("mlp", {}) if isinstance(cfg, dict) else {}
    return {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None",
            "n_neurons": int(m.get("n_neurons", hidden_dim)), "n_hidden_layers": int(m.get("n_hidden_layers", 1)),
            "dtype": str(m.get("dtype", "fp32"))}          # "fp16": tcnn's operand precision (BASELINE configs[4])


def fused_cat(pe, features):
    """``torch.cat((pe, features), -1)`` without the copy when both are the adjacent column views that
    ``Pos_Encoding.forward`` returned."""
    b = pe._base
    if (b is not None and b is features._base and b.dim() == 2 and pe.dim() == 2 and features.dim() == 2
            and pe.storage_offset() == b.storage_offset()
            and features.storage_offset() == b.storage_offset() + pe.shape[1]
            and pe.shape[1] + features.shape[1] == b.shape[1] and pe.stride() == b.stride()
            and features.stride() == b.stride()):
        return b
    return torch.cat((pe, features), -1)


class Decoder(nn.Module):
    """reference models/decoder.py:7-27."""

    def __init__(self, cfg, bound, n_class=40):
        super().__init__()
        self.pe_fn = Pos_Encoding(cfg, bound)
        self.pe_dim = self.pe_fn.pe_dim
        self.grid_dim = self.pe_fn.grid_dim
        self.pts_dim = cfg["pts_dim"]
        self.hidden_dim = cfg["hidden_dim"]
        self.pixel_dim = cfg["pixel_dim"]
        self.n_class = n_class
        mc = _mlp_cfg(cfg, self.hidden_dim)
        self.coarse_fn = Coarse(pts_dim=self.pe_dim, hidden_dim=self.hidden_dim, feature_dim=self.grid_dim, net_cfg=mc)
        self.out_fn = Out(pts_dim=self.pe_dim, feature_dim=self.hidden_dim * 2, hidden_dim=self.hidden_dim,
                          n_class=self.n_class, net_cfg=mc)
        self.merge = Merge(cfg, hidden_dim=self.hidden_dim, feature_dim=self.pixel_dim, bound=bound, net_cfg=mc)


class Pos_Encoding(nn.Module):
    """reference models/decoder.py:30-48."""

    def __init__(self, cfg, bound):
        super().__init__()
        self.pe_fn, self.pe_dim = get_encoder(cfg["pos"]["method"], n_bins=cfg["pos"]["n_bins"])
        dim_max = (bound[:, 1] - bound[:, 0]).max()
        self.resolution = int(dim_max / cfg["grid"]["voxel_size"])
        self.grid_fn, self.grid_dim = get_encoder(cfg["grid"]["method"], log2_hashmap_size=cfg["grid"]["hash_size"],
                                                  desired_resolution=self.resolution)

    def forward(self, pts):
        buf = ops.encode(pts.float(), self.grid_fn.params, self.grid_fn.meta, None, self.pe_fn.n_bins, True, True)
        return buf[:, :self.pe_dim], buf[:, self.pe_dim:]

    def forward_world(self, pts_world, bound):
        """Fused fp64 normalisation (slams/mapping.py:608) + encoding; returns the [P, 80] buffer."""
        return ops.encode(pts_world, self.grid_fn.params, self.grid_fn.meta, bound, self.pe_fn.n_bins, True, True)


class Merge(nn.Module):
    """reference models/decoder.py:51-77 (2-D feature branch)."""

    def __init__(self, cfg, hidden_dim=32, feature_dim=64, bound=None, net_cfg=None):
        super().__init__()
        self.bound = bound
        self.pe_fn, self.pe_dim = get_encoder(cfg["pos"]["method"], n_bins=cfg["pos"]["n_bins"])
        self.decoder = tcnn.Network(n_input_dims=self.pe_dim + feature_dim, n_output_dims=hidden_dim,
                                    network_config=net_cfg or _mlp_cfg(cfg, hidden_dim))

    def forward(self, p, o, features=None):
        n_refer, n_points, C = features.shape
        # (p - b0)/(b1 - b0) in fp64 (the RELATIVE vector normalised with the scene bound) fused with OneBlob
        pe = ops.encode(p.flatten(0, 1), None, None, self.bound, self.pe_fn.n_bins, True, False)
        latents = self.decoder(torch.cat((pe, features.flatten(0, 1)), -1))
        return torch.mean(latents.reshape(n_refer, n_points, -1), 0)


class Coarse(nn.Module):
    """reference models/decoder.py:80-94."""

    def __init__(self, pts_dim, hidden_dim, feature_dim, net_cfg=None):
        super().__init__()
        self.decoder = tcnn.Network(n_input_dims=pts_dim + feature_dim, n_output_dims=hidden_dim + 1,
                                    network_config=net_cfg or _mlp_cfg({}, hidden_dim))

    def forward(self, pe, features=None):
        return self.decoder(fused_cat(pe, features)).float()


class Out(nn.Module):
    """reference models/decoder.py:97-125."""

    def __init__(self, pts_dim, feature_dim, hidden_dim, n_class, net_cfg=None):
        super().__init__()
        cfg = net_cfg or _mlp_cfg({}, hidden_dim)
        self.color_decoder = tcnn.Network(n_input_dims=pts_dim + feature_dim, n_output_dims=3, network_config=cfg)
        self.logit_decoder = tcnn.Network(n_input_dims=pts_dim + feature_dim, n_output_dims=n_class, network_config=cfg)
        self.sigmoid = nn.Sigmoid()

    def forward(self, pe, features):
        # the reference builds cat((pe, features)) twice (:123-124); here both networks read the two column blocks in place
        c, l = self.color_decoder, self.logit_decoder
        color = self.sigmoid(ops.mlp_cat(pe, features, c.params, c.n_output_dims, c.n_neurons, c.n_hidden_layers, c.fp16))
        logit = ops.mlp_cat(pe, features, l.params, l.n_output_dims, l.n_neurons, l.n_hidden_layers, l.fp16)
        return color, logit
