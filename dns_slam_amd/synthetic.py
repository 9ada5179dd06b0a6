"""Synthetic RGB-D + semantics frames of an analytic box room and the reference's config keys.

Replica / ScanNet are not available offline; benchmarks and tests use this scene instead: room_0's bound
(reference configs/replica/room_0.yaml:4) enlarged by the 0.32 rule (slams/dns_slam.py:100-107), a pinhole camera,
K target frames on a circle looking outward, depth = exact ray / wall intersection (z-depth along the un-normalised
ray, the reference's convention), 2 % zero-depth pixels, colour = 0.5 + 0.5 sin(k.p) of the hit point, 8 labels from
the 1-metre checker cell of the hit point.  Plain CPU torch, deterministic per seed.
"""
from __future__ import annotations

import copy
import math

import torch

ROOM0_BOUND = [[-2.9, 8.9], [-3.2, 5.5], [-3.5, 3.3]]
SCENE0000_BOUND = [[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]]
OFFICE0_BOUND = [[-5.5, 5.9], [-6.7, 5.4], [-4.7, 5.3]]      # reference configs/replica/office_0.yaml:4


def load_bound(bound, bound_divisible=0.32, scale=1.0) -> torch.Tensor:
    """reference slams/dns_slam.py:100-107 (float64)."""
    b = torch.tensor(bound, dtype=torch.float64) * scale
    b[:, 1] = (((b[:, 1] - b[:, 0]) / bound_divisible).int() + 1) * bound_divisible + b[:, 0]
    return b


def default_cfg(n_pixels=2000, n_samples_ray=32, n_surface_ray=15, n_frames=4, hash_size=16, voxel_size=0.02,
                n_neurons=32, n_hidden_layers=1, smooth_pts=64, track_pixels=500, track_iters=50, mlp_dtype="fp32") -> dict:
    """The keys of reference configs/slam.yaml + configs/replica/replica.yaml that shape the hot path."""
    return {
        "model": {"pts_dim": 3, "pixel_dim": 64, "hidden_dim": 32,
                  "pos": {"method": "OneBlob", "n_bins": 16},
                  "grid": {"method": "HashGrid", "hash_size": hash_size, "voxel_size": voxel_size},
                  "mlp": {"n_neurons": n_neurons, "n_hidden_layers": n_hidden_layers, "dtype": mlp_dtype}},
        "training": {"lr": 0.005, "lambda_color": 5.0, "lambda_depth": 5.0, "lambda_label": 0.1,
                     "lambda_smooth": 0.00001, "lambda_fs": 10, "lambda_opacity": 10,
                     "n_samples_ray": n_samples_ray, "n_surface_ray": n_surface_ray, "smooth_pts": smooth_pts,
                     "opacity_sigma": 0.05},
        "tracking": {"cam_lr": 0.001, "n_iters": track_iters, "n_pixels": track_pixels},
        "mapping": {"BA_cam_lr": 0.0005, "start_optimize_idx": 10, "n_joint_optimize_frames": n_frames,
                    "n_pixels": n_pixels, "n_iters": 100},
        "seperate_LR": False,
    }


def camera(H=480, W=640, fx=500.0, fy=500.0):
    return {"H": H, "W": W, "fx": fx, "fy": fy, "cx": (W - 1) / 2.0, "cy": (H - 1) / 2.0}


def make_pose(theta: float, center=(3.0, 1.2, 0.0), radius=1.0, jitter: torch.Tensor = None) -> torch.Tensor:  # noqa: E501
    """c2w of a camera on a horizontal circle looking outward (camera looks along -z, y up; world z up)."""
    look = torch.tensor([math.cos(theta), math.sin(theta), 0.0], dtype=torch.float64)
    if jitter is not None:
        look = look + 0.05 * jitter[:3]
        look = look / look.norm()
    up = torch.tensor([0.0, 0.0, 1.0], dtype=torch.float64)
    x = torch.linalg.cross(look, up)
    x = x / x.norm()
    # right-handed frame: x right, y up, z = -look
    z = -look
    y = torch.linalg.cross(z, x)
    c2w = torch.eye(4, dtype=torch.float64)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2] = x, y, z
    c2w[:3, 3] = torch.tensor(center, dtype=torch.float64) + radius * torch.tensor([math.cos(theta), math.sin(theta), 0.0],
                                                                                   dtype=torch.float64)
    return c2w.float()


def render_frame(c2w: torch.Tensor, cam: dict, room: torch.Tensor, seed: int, zero_frac=0.02):
    """Exact RGB-D-label images of the box ``room`` [3,2] seen from ``c2w``."""
    H, W = cam["H"], cam["W"]
    g = torch.Generator().manual_seed(seed)
    jj, ii = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    dirs = torch.stack([(ii - cam["cx"]) / cam["fx"], -(jj - cam["cy"]) / cam["fy"], -torch.ones_like(ii)], -1)
    R = c2w[:3, :3].double()
    o = c2w[:3, 3].double()
    d = (dirs[..., None, :] * R).sum(-1)                                  # [H,W,3]
    t = (room[None, None] - o[None, None, :, None]) / d[..., None]        # [H,W,3,2]
    tmax, side = t.max(-1)
    depth, axis = tmax.min(-1)                                            # exit distance; which axis' wall
    hit = o + d * depth[..., None]
    # 8 labels: 1-metre checker cells of the hit point (every view sees all classes, patches of ~100 px)
    cell = torch.floor(hit + 1e-6).long()
    label = (cell[..., 0] + 3 * cell[..., 1] + 5 * cell[..., 2]) % 8
    k = torch.tensor([[1.3, 2.1, 0.7], [0.9, 1.7, 2.3], [2.2, 0.6, 1.1]], dtype=torch.float64)
    color = 0.5 + 0.5 * torch.sin(hit @ k.t())
    depth = depth.float()
    depth[torch.rand(H, W, generator=g) < zero_frac] = 0.0
    return color.float(), depth, label.float()


def make_scene(n_frames=4, cam=None, bound=None, seed=0):
    """-> (bound fp64 [3,2], cam dict, target_frames dict as Mapper.optimize expects)."""
    cam = cam or camera()
    bound_t = load_bound(bound or ROOM0_BOUND)
    room = bound_t.clone()
    room[:, 0] += 0.6
    room[:, 1] -= 0.6                                                     # walls strictly inside the bound
    g = torch.Generator().manual_seed(seed)
    colors, depths, labels, poses = [], [], [], []
    for f in range(n_frames):
        jit = torch.randn(3, generator=g, dtype=torch.float64)
        centre = tuple(float(v) for v in (bound_t[:, 0] + bound_t[:, 1]) / 2)          # camera circle around the room centre
        c2w = make_pose(2 * math.pi * f / n_frames + 0.3, center=centre, radius=min(1.0, float((bound_t[:, 1] - bound_t[:, 0]).min()) / 8), jitter=jit)
        c, d, l = render_frame(c2w, cam, room, seed * 100 + f)
        colors.append(c), depths.append(d), labels.append(l), poses.append(c2w)
    label_dict = sorted({int(v) for l in labels for v in torch.unique(l).tolist()})
    frames = {"gt_color": torch.stack(colors), "gt_depth": torch.stack(depths), "gt_label": torch.stack(labels),
              "est_c2w": torch.stack(poses), "gt_c2w": torch.stack(poses), "label_dict": label_dict}
    return bound_t, cam, frames


def clone_cfg(cfg):
    return copy.deepcopy(cfg)
