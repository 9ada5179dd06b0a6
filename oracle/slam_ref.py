"""Oracle: restatement of the reference's optimise-step bodies.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``slams/mapping.py`` and
``slams/tracking.py`` cannot be imported in the build container (they need cv2,
colorama, tinycudann), so this file restates, line by cited line, the tensor-level
bodies of ``Mapper.renderer`` (mapping.py:603-635), ``Mapper.fine_fn`` (:590-601),
``Mapper.get_target_samples`` (:471-588, without the 2-D feature branch, whose output
``features`` is an input here), ``Mapper.smoothness`` (:129-159), the mapping losses
(:891-907), ``Tracker.renderer`` (tracking.py:188-214), ``Tracker.get_target_samples``
(:128-186) and the tracking losses (:326-329) on top of ``render_math`` (pinned to the
imported ``utils/common.py``) and ``tcnn_ref`` (parity unpinned).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from . import render_math as rm
from . import tcnn_ref as tr


@dataclass
class ModelCfg:
    """The keys of reference ``configs/slam.yaml:15-26`` that shape the path, plus the MLP
    shape (reference: 1 hidden layer of 32; BASELINE config 2: 2 x 64)."""
    n_bins: int = 16
    hash_size: int = 16
    voxel_size: float = 0.02
    hidden_dim: int = 32          # latent width (decoder.py:15), also tcnn n_neurons in the reference
    n_neurons: int = 32
    n_hidden_layers: int = 1
    pixel_dim: int = 64
    n_class: int = 40
    n_levels: int = 16
    level_dim: int = 2
    base_resolution: int = 16


class OracleModel:
    """Parameter container mirroring ``Decoder`` (models/decoder.py:7-27) + the Mapper's
    ``fine_decoders`` dict (slams/mapping.py:736-749).  All parameters are flat fp32 leaf
    tensors with ``requires_grad``."""

    def __init__(self, cfg: ModelCfg, bound: torch.Tensor, seed_table: int = 3, seed_mlp: int = 4,
                 fine_classes=()):
        self.cfg = cfg
        self.bound = bound.double()
        self.resolution = tr.desired_resolution(self.bound, cfg.voxel_size)
        self.meta = tr.grid_meta(cfg.hash_size, self.resolution, cfg.n_levels, cfg.level_dim, cfg.base_resolution)
        self.pe_dim = 3 * cfg.n_bins
        self.grid_dim = cfg.n_levels * cfg.level_dim
        g = torch.Generator().manual_seed(seed_table)
        self.table = tr.grid_init(self.meta, g).requires_grad_(True)
        g = torch.Generator().manual_seed(seed_mlp)
        in80 = self.pe_dim + self.grid_dim
        in112 = self.pe_dim + 2 * cfg.hidden_dim
        self.shapes = {
            "coarse": (in80, cfg.hidden_dim + 1),
            "color": (in112, 3),
            "logit": (in112, cfg.n_class),
            "fine": (in80, cfg.hidden_dim + 1),
        }
        mk = lambda k: tr.mlp_init(self.shapes[k][0], self.shapes[k][1], cfg.n_neurons, cfg.n_hidden_layers, g).requires_grad_(True)
        self.coarse = mk("coarse")
        self.color = mk("color")
        self.logit = mk("logit")
        self.fine: Dict[int, torch.Tensor] = {int(c): mk("fine") for c in fine_classes}
        self.taps = None                         # set to {} to record x and dL/d(grid features) of every pe_fn call (tests)

    def parameters(self):
        return [self.table, self.coarse, self.color, self.logit] + [self.fine[k] for k in sorted(self.fine)]

    def zero_grad(self):
        for p in self.parameters():
            p.grad = None

    # models/decoder.py:45-48
    def pe_fn(self, x):
        x = x.float()
        grid = tr.hashgrid_forward(x, self.table, self.meta)
        if self.taps is not None and grid.requires_grad:
            i = len(self.taps)
            self.taps[i] = {"x": x.detach()}
            grid.register_hook(lambda g, i=i: self.taps[i].__setitem__("d_grid", g.detach()))
        return tr.oneblob_forward(x, self.cfg.n_bins), grid

    def _mlp(self, name, params, x):
        n_in, n_out = self.shapes[name]
        return tr.mlp_forward(x, params, n_in, n_out, self.cfg.n_neurons, self.cfg.n_hidden_layers)

    # models/decoder.py:93-94
    def coarse_fn(self, pe, features):
        return self._mlp("coarse", self.coarse, torch.cat((pe, features), -1))

    # models/decoder.py:122-125
    def out_fn(self, pe, features):
        x = torch.cat((pe, features), -1)
        return torch.sigmoid(self._mlp("color", self.color, x)), self._mlp("logit", self.logit, x)

    # slams/mapping.py:590-601
    def fine_fn(self, pes, classes, features):
        latents = torch.zeros(pes.shape[0], self.cfg.hidden_dim + 1)
        for c in torch.unique(classes, sorted=True).tolist():
            if c not in self.fine:
                raise ValueError("Fine decoders does NOT have class", c)
            index = classes == c
            if int(index.sum()) > 1:
                y = self._mlp("fine", self.fine[c], torch.cat((pes[index], features[index]), -1))
                latents = latents.index_put((torch.nonzero(index).reshape(-1),), y)
        return latents


def point_classes(gt_label: torch.Tensor, n_samples: int, layout: str = "reference_tiled"):
    """slams/mapping.py:612-613: ``gt_label.repeat(1, n_samples).flatten(0, 1)`` on a 1-D
    [N] tensor TILES the labels, so ray-major point k gets label[k mod N].
    ``per_ray`` is the evidently intended label[k // S]."""
    if layout == "reference_tiled":
        return gt_label.repeat(1, n_samples).flatten(0, 1)
    if layout == "per_ray":
        return gt_label.repeat_interleave(n_samples)
    raise ValueError(layout)


def mapper_renderer(model: OracleModel, samples: dict, label_layout: str = "reference_tiled"):
    """slams/mapping.py:603-635 -> (color[N,3], depth[N], var[N], logits[N,C], fine[P,33], coarse[P,33])."""
    pts = samples["pts"]
    n_pts, n_samples, _ = pts.shape
    x = rm.normalise_points(pts.flatten(0, 1), model.bound)
    z_vals = samples["z_vals"]
    classes = point_classes(samples["gt_label"], n_samples, label_layout)
    pixel_pts = samples["features"].flatten(0, 1)
    pe, grid = model.pe_fn(x)
    coarse = model.coarse_fn(pe, grid)
    fine = model.fine_fn(pe, classes, grid)
    color, logits = model.out_fn(pe, torch.cat((fine[:, 1:], pixel_pts), -1))
    values = torch.cat((color, fine[:, 0:1]), -1).reshape(n_pts, n_samples, -1)
    logits = logits.reshape(n_pts, n_samples, -1)
    depth, var, rgb, weights = rm.raw2nerf_color(values, z_vals)
    pred_logits = torch.sum(weights[..., None] * logits, -2)
    return rgb, depth, var, pred_logits, fine, coarse


def tracker_renderer(model: OracleModel, samples: dict):
    """slams/tracking.py:188-214 (coarse-only) -> (color, depth, var, logits)."""
    pts = samples["pts"].flatten(0, 1)
    x = rm.normalise_points(pts, model.bound)
    z_vals = samples["z_vals"]
    n_pts, n_samples = z_vals.shape
    pixel_pts = samples["features"].flatten(0, 1)
    pe, grid = model.pe_fn(x)
    lat = model.coarse_fn(pe, grid)
    color, logits = model.out_fn(pe, torch.cat((lat[:, 1:], pixel_pts), -1))
    values = torch.cat((color, lat[:, 0:1]), -1).reshape(n_pts, n_samples, -1)
    logits = logits.reshape(n_pts, n_samples, -1)
    depth, var, rgb, weights = rm.raw2nerf_color(values, z_vals)
    pred_logits = torch.sum(weights[..., None] * logits, -2)
    return rgb, depth, var, pred_logits


def eval_points(model: OracleModel, pts: torch.Tensor, pixel_pts: torch.Tensor, gt_label_pts=None, stage: str = "fine"):
    """slams/meshing.py:461-503 (``Mesher.eval_points``, the live branch): colour + occupancy logit of arbitrary world
    points, -100 occupancy outside the (open) bound; ``stage='fine'`` routes every point through the fine decoder of
    its label (meshing.py:447-458, same > 1 point rule as Mapper.fine_fn) and also returns argmax labels (-1 outside)."""
    b = model.bound
    mask = ((pts[:, 0] < b[0, 1]) & (pts[:, 0] > b[0, 0]) & (pts[:, 1] < b[1, 1]) & (pts[:, 1] > b[1, 0]) &
            (pts[:, 2] < b[2, 1]) & (pts[:, 2] > b[2, 0]))
    x = rm.normalise_points(pts, b)
    pe, grid = model.pe_fn(x)
    if stage == "coarse":
        lat = model.coarse_fn(pe, grid)
    else:
        lat = model.fine_fn(pe, gt_label_pts, grid)
    color, logits = model.out_fn(pe, torch.cat((lat[:, 1:], pixel_pts), -1))
    values = torch.cat((color, lat[:, 0:1]), -1)
    values[~mask, 3] = -100
    if stage == "coarse":
        return values, None
    labels = torch.argmax(logits, dim=-1)
    labels[~mask] = -1
    return values, labels


def frame_samples(image5, quad, T, cam, bound, indices, t_surf, t_zero, n_samples, n_surface,
                  window=None, features=None, pixel_dim_half=32):
    """One target frame of ``get_target_samples`` (slams/mapping.py:487-572 /
    slams/tracking.py:128-186) with the pixel indices and jitter explicit.
    ``image5`` = cat(color, depth, label) [H,W,5]; cam = (H,W,fx,fy,cx,cy)."""
    H, W, fx, fy, cx, cy = cam
    H0, H1, W0, W1 = window if window is not None else (0, H, 0, W)
    R = rm.rotation_from_quad(quad)
    i, j = rm.uv_from_indices(indices, H0, H1, W0, W1)
    px = rm.gather_pixels(indices, image5, H0, H1, W0, W1)
    rays_o, rays_d = rm.rays_from_uv(i, j, R, T, fx, fy, cx, cy)
    gt_color, gt_depth, gt_label = px[:, :3], px[:, 3], px[:, 4]
    with torch.no_grad():
        far_bb, inside = rm.box_far(rays_o, rays_d, gt_depth, bound)
    z = rm.sample_along_rays(gt_depth, n_samples, n_surface, far_bb, t_surf, t_zero)
    pts = rm.points_from_rays(rays_o, rays_d, z)
    if features is None:
        features = torch.zeros(pts.shape[0], pts.shape[1], pixel_dim_half)
    code = features * rm.truncation_mask(z, gt_depth)[..., None]
    return {"gt_color": gt_color.float(), "gt_depth": gt_depth.float(), "gt_label": gt_label.to(torch.int64),
            "rays_o": rays_o.float(), "rays_d": rays_d.float(), "pts": pts.float(), "z_vals": z.float(),
            "inside": inside, "features": code}


def mapper_target_samples(frames: list):
    """Concatenate per-frame dicts and drop rays leaving the box (slams/mapping.py:564-586)."""
    keys = ["gt_color", "gt_depth", "gt_label", "rays_o", "rays_d", "pts", "z_vals", "features"]
    mask = torch.cat([f["inside"] for f in frames], 0)
    return {k: torch.cat([f[k] for f in frames], 0)[mask] for k in keys}


def smoothness(model: OracleModel, sample_points: int, u_offset: torch.Tensor, u_jitter: torch.Tensor,
               voxel_size: float = 0.1, margin: float = 0.05):
    """slams/mapping.py:129-159 with the two CPU draws explicit (``torch.rand(3)`` then
    ``torch.rand((1,1,1,3))``); :152's ``reshape(pts_shape[:3], 1)`` is restated as the
    evidently intended ``reshape(*pts_shape[:3], 1)``."""
    bound = model.bound
    volume = bound[:, 1] - bound[:, 0]
    grid_size = (sample_points - 1) * voxel_size
    offset_max = bound[:, 1] - bound[:, 0] - grid_size - 2 * margin
    offset = u_offset.to(offset_max) * offset_max + margin
    n = sample_points - 1
    ar = torch.arange(0, n, dtype=torch.long)
    gx, gy, gz = torch.meshgrid(ar, ar, ar, indexing="ij")
    coords = torch.stack([gx, gy, gz], dim=-1).float().to(volume)
    pts = (coords + u_jitter.to(volume)) * voxel_size + bound[:, 0] + offset
    pts = (pts - bound[:, 0]) / (bound[:, 1] - bound[:, 0])
    shp = pts.shape
    pe, grid = model.pe_fn(pts.reshape(-1, 3))
    occ = model.coarse_fn(pe, grid)[:, 0:1].reshape(*shp[:3], 1)
    tv_x = torch.pow(occ[1:, ...] - occ[:-1, ...], 2).sum()
    tv_y = torch.pow(occ[:, 1:, ...] - occ[:, :-1, ...], 2).sum()
    tv_z = torch.pow(occ[:, :, 1:, ...] - occ[:, :, :-1, ...], 2).sum()
    return (tv_x + tv_y + tv_z) / (sample_points ** 3)


@dataclass
class LossCfg:
    """reference ``configs/replica/replica.yaml:20-31``."""
    lambda_p: float = 5.0
    lambda_d: float = 5.0
    lambda_l: float = 0.1
    lambda_lt: float = 10.0
    lambda_sm: float = 1e-5
    lambda_fs: float = 10.0
    lambda_opacity: float = 10.0
    opacity_sigma: float = 0.05
    smooth_pts: int = 64


def mapping_loss(model: OracleModel, samples: dict, lc: LossCfg, u_offset=None, u_jitter=None,
                 label_layout="reference_tiled"):
    """The seven-term loss of one mapping iteration, slams/mapping.py:887-907."""
    rgb, depth, var, logits, fine, coarse = mapper_renderer(model, samples, label_layout)
    terms = {
        "d": rm.depth_loss(samples["gt_depth"], depth),
        "p": rm.photometric_loss(samples["gt_color"], rgb),
        "l": rm.label_loss(samples["gt_label"], logits),
        "lt": rm.latent_loss(coarse, fine),
    }
    fs, op = rm.opacity_loss(samples["z_vals"], samples["gt_depth"], fine[..., -1], lc.opacity_sigma)
    terms["fs"], terms["op"] = fs, op
    loss = lc.lambda_p * terms["p"] + lc.lambda_d * terms["d"] + lc.lambda_l * terms["l"] + lc.lambda_lt * terms["lt"] \
        + lc.lambda_fs * fs + lc.lambda_opacity * op
    if u_offset is not None:
        terms["sm"] = smoothness(model, lc.smooth_pts, u_offset, u_jitter)
        loss = loss + lc.lambda_sm * terms["sm"]
    outs = {"rgb": rgb, "depth": depth, "var": var, "logits": logits, "fine": fine, "coarse": coarse}
    return loss, terms, outs


def tracking_loss(model: OracleModel, samples: dict, mask: torch.Tensor, lambda_p=5.0, lambda_d=5.0, lambda_l=0.1):
    """slams/tracking.py:323-329."""
    rgb, depth, var, logits = tracker_renderer(model, samples)
    p = rm.track_photometric_loss(samples["gt_color"], rgb, mask)
    d = rm.track_depth_loss(samples["gt_depth"], depth, var, mask)
    l = rm.track_label_loss(samples["gt_label"], logits, mask)
    return lambda_p * p + lambda_d * d + lambda_l * l, {"p": p, "d": d, "l": l}, \
        {"rgb": rgb, "depth": depth, "var": var, "logits": logits}
