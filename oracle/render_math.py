"""Oracle: restatement of the live render math of reference ``utils/common.py``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Plain PyTorch on CPU.  Every
function cites the reference lines it follows.  Random draws are explicit
arguments wherever the reference draws them internally, so that the HIP path
and this oracle can be fed identical integers / uniforms; ``*_replay`` helpers
draw them from the global torch generator in the reference's call order
 so the imported reference can be reproduced exactly.

Pinned against the imported reference by ``tests/golden/*.npz``
(``tests/test_oracle_golden.py``).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- a1
def uv_from_indices(indices: torch.Tensor, H0: int, H1: int, W0: int, W1: int):
    """Flat window index -> (i=col, j=row) as float32.

    Reference ``get_sample_uv`` (utils/common.py:282-293) builds two meshgrids of
    ``linspace`` (step exactly 1.0) over the window, transposes them and indexes
    the flattened result (``select_uv`` :266-280): flat q -> row H0 + q // (W1-W0),
    col W0 + q % (W1-W0).
    """
    w = W1 - W0
    j = (H0 + torch.div(indices, w, rounding_mode="floor")).to(torch.float32)
    i = (W0 + indices % w).to(torch.float32)
    return i, j


def gather_pixels(indices: torch.Tensor, image: torch.Tensor, H0: int, H1: int, W0: int, W1: int):
    """``color[H0:H1, W0:W1].reshape(-1, C)[indices]`` (utils/common.py:277-279,287)."""
    c = image.shape[-1]
    return image[H0:H1, W0:W1].reshape(-1, c)[indices]


def class_balanced_indices(label_window: torch.Tensor, n: int, draw=None):
    """Flat indices of the class-balanced pick (utils/common.py:307-330).

    ``label_window`` is the ``[H1-H0, W1-W0]`` label plane.  ``draw(k, m)`` returns
    ``m`` integers in ``[0, k)``; default ``torch.randint`` on the global generator,
    i.e. the reference's own call order (one draw per class, ascending class id;
    the first class gets the remainder :318-321; a single-pixel class is repeated
    :324-325).
    """
    if draw is None:
        draw = lambda k, m: torch.randint(k, (m,))
    flat = label_window.reshape(-1)
    classes = torch.unique(flat, sorted=True)
    n_class = classes.numel()
    n_k = n // n_class
    out = []
    for c in range(n_class):
        m = n - n_k * (n_class - 1) if c == 0 else n_k
        idx_c = torch.nonzero(flat == classes[c]).reshape(-1)
        k = idx_c.numel()
        if k == 1:
            out.append(idx_c.repeat(m))
        else:
            out.append(idx_c[draw(k, m)])
    return torch.cat(out, dim=-1)


# --------------------------------------------------------------------------- a4
def quad2rotation(quad: torch.Tensor) -> torch.Tensor:
    """Quaternion (w,x,y,z) [B,4] -> R [B,3,3] with two_s = 2/|q|^2, no normalise.

    utils/common.py:406-429 (the reference allocates with ``.to(quad.get_device())``
    and therefore raises on CPU tensors -- SURVEY D8; the formula is restated on
    ``quad.device``).
    """
    qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    two_s = 2.0 / (quad * quad).sum(-1)
    rows = [
        1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
        two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr),
        two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2),
    ]
    return torch.stack(rows, -1).reshape(-1, 3, 3)


def rotation_from_quad(quad: torch.Tensor) -> torch.Tensor:
    """utils/common.py:447-458 (1-D input -> [3,3])."""
    if quad.dim() == 1:
        return quad2rotation(quad[None])[0]
    return quad2rotation(quad)


# --------------------------------------------------------------------------- a3
def rays_from_uv(i, j, R, T, fx, fy, cx, cy):
    """utils/common.py:248-264: dir=((i-cx)/fx, -(j-cy)/fy, -1); d = sum(dir*R, -1); o = T."""
    dirs = torch.stack([(i - cx) / fx, -(j - cy) / fy, -torch.ones_like(i)], -1)
    dirs = dirs.reshape(-1, 1, 3)
    rays_d = torch.sum(dirs * R, -1)
    rays_o = T.expand(rays_d.shape)
    return rays_o, rays_d


# --------------------------------------------------------------------------- a5
def box_far(rays_o, rays_d, gt_depth, bound):
    """Box far clip, inline in slams/mapping.py:519-527 (= :653-660, :792-800,
    slams/tracking.py:148-156).  ``bound`` is float64 [3,2] so everything promotes to
    fp64.  Returns (far_bb [n,1] fp64 with the +0.01 applied, inside_mask [n] bool
    evaluated BEFORE the +0.01)."""
    det_o = rays_o.detach().unsqueeze(-1)
    det_d = rays_d.detach().unsqueeze(-1)
    t = (bound.unsqueeze(0) - det_o) / det_d
    far_bb, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
    inside = far_bb >= gt_depth
    far_bb = far_bb.unsqueeze(-1) + 0.01
    return far_bb, inside


# --------------------------------------------------------------------------- a6
def surface_jitter_replay(n_surface: int):
    """The two ``torch.rand(n_surface)`` draws of sample_along_rays, in order
    (utils/common.py:571,582), including the forced 0.5 (:572-573)."""
    t = torch.rand(n_surface)
    if not torch.any(t == 0.5):
        t[n_surface // 2 + 1] = 0.5
    t0 = torch.rand(n_surface)
    return t, t0


def sample_along_rays(gt_depth, n_samples, n_surface, far_bb, t_surf, t_zero):
    """Depth-guided z sampling, utils/common.py:561-599, with the two jitter vectors
    explicit (``t_surf`` already carries the forced 0.5).

    d>0 rays: n_surface values 0.95 d (1-t) + 1.05 d t (fp32); d==0 rays: one shared
    row 0.001 (1-t') + max(d) t' (fp32, :580-585); n_samples uniform values between
    0.001 d (fp32) and clamp(far_bb, 0, max(1.2 d)) (fp64) on linspace(0,1) (:589-593);
    concatenated, sorted ascending per ray (in fp64), cast to fp32 (:595-599).
    """
    gt_depth = gt_depth.reshape(-1, 1)
    nz = (gt_depth > 0).squeeze(-1)
    d_nz = gt_depth[nz].reshape(-1, 1).repeat(1, n_surface)
    z_nz = 0.95 * d_nz * (1.0 - t_surf) + 1.05 * d_nz * t_surf
    z_near = torch.zeros(gt_depth.shape[0], n_surface)
    z_near[nz, :] = z_nz
    near = 0.001
    far = torch.max(gt_depth)
    z_zero = near * (1.0 - t_zero) + far * t_zero
    z_near[~nz, :] = z_zero
    if n_samples > 0:
        d_rep = gt_depth.repeat(1, n_samples)
        near_u = d_rep * 0.001
        far_u = torch.clamp(far_bb, 0, torch.max(gt_depth * 1.2))
        t_vals = torch.linspace(0.0, 1.0, steps=n_samples)
        z_u = near_u * (1.0 - t_vals) + far_u * t_vals
        z, _ = torch.sort(torch.cat([z_u, z_near], -1), -1)
    else:
        z, _ = torch.sort(z_near, -1)
    return z.float()


# --------------------------------------------------------------------------- a7
def points_from_rays(rays_o, rays_d, z_vals):
    """slams/mapping.py:531, slams/tracking.py:160."""
    return rays_o[:, None, :] + rays_d[:, None, :] * z_vals[:, :, None]


def normalise_points(pts, bound):
    """slams/mapping.py:608, slams/tracking.py:190: fp64 promotion through ``bound``."""
    return (pts - bound[:, 0]) / (bound[:, 1] - bound[:, 0])


def truncation_mask(z_vals, gt_depth):
    """slams/mapping.py:553-556 / slams/tracking.py:165-168 (multiplies the 2-D code)."""
    d = gt_depth[:, None]
    front = torch.where(z_vals < d * 0.95, torch.ones_like(z_vals), torch.zeros_like(z_vals))
    back = torch.where(z_vals > d * 1.05, torch.ones_like(z_vals), torch.zeros_like(z_vals))
    dm = torch.where(d > 0.0, torch.ones_like(d), torch.zeros_like(d))
    return (1.0 - front) * (1.0 - back) * dm


# -------------------------------------------------------------------------- a13
def raw2nerf_color(raw, z_vals):
    """Occupancy compositing, utils/common.py:506-537 with ``occupancy=True`` (the only
    mode the reference ever uses; ``dists``/``rays_d`` are dead in that mode).

    alpha = sigmoid(10 raw[...,3]); w = alpha * exclusive_cumprod(1 - alpha + 1e-10);
    w /= sum(w) (no epsilon, D9); depth = sum w z; var = sum w (z-depth)^2; rgb = sum w c.
    """
    rgb = raw[..., :3]
    alpha = torch.sigmoid(10 * raw[..., -1])
    ones = torch.ones((alpha.shape[0], 1), dtype=alpha.dtype)
    trans = torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-10], -1), -1)[..., :-1]
    weights = alpha * trans
    weights = weights / weights.sum(dim=-1)[:, None]
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    tmp = z_vals - depth_map.unsqueeze(-1)
    depth_var = torch.sum(weights * tmp * tmp, dim=-1)
    return depth_map, depth_var, rgb_map, weights


# -------------------------------------------------------------------------- a14
def opacity_loss(z_vals, depth, occ, truncation=0.2, sigma=0.05):
    """utils/common.py:769-802.  NB the reference call sites pass ``opacity_sigma`` into
    ``truncation`` and leave ``sigma`` at 0.05, and pass the LAST latent channel as
    ``occ`` (slams/mapping.py:896) -- callers of this oracle do the same."""
    bs, n_sample = z_vals.shape
    depth = depth.unsqueeze(-1)
    occ = torch.sigmoid(10 * occ).reshape(bs, n_sample)
    front = torch.where(z_vals < (depth - truncation), torch.ones_like(z_vals), torch.zeros_like(z_vals))
    back = torch.where(z_vals > (depth + truncation), torch.ones_like(z_vals), torch.zeros_like(z_vals))
    dmask = torch.where(depth > 0.0, torch.ones_like(depth), torch.zeros_like(depth))
    omask = (1.0 - front) * (1.0 - back) * dmask
    if torch.count_nonzero(front) > 0 and torch.count_nonzero(omask) > 0:
        fs = ((occ * front * dmask) ** 2).mean()
        pseudo = 0.5 * torch.exp(-0.5 * ((z_vals - depth) / sigma) ** 2)
        op = ((occ * omask - pseudo * omask) ** 2).mean()
    else:
        fs = torch.tensor(0)
        op = torch.tensor(0)
    return fs, op


def photometric_loss(gt_color, pred_color):
    """slams/mapping.py:110-112."""
    return ((gt_color - pred_color) ** 2).mean()


def depth_loss(gt_depth, pred_depth):
    """slams/mapping.py:114-117 (mean |d - d^| over d > 0)."""
    m = gt_depth > 0
    return torch.abs(gt_depth[m] - pred_depth[m]).mean()


def label_loss(gt_label, pred_logits):
    """slams/mapping.py:119-121."""
    return F.cross_entropy(pred_logits, gt_label)


def latent_loss(coarse, fine):
    """slams/mapping.py:123-126."""
    return ((coarse - fine) ** 2).mean()


# -------------------------------------------------------------------------- a15
def track_photometric_loss(gt_color, pred_color, mask):
    """slams/tracking.py:85-87."""
    return ((gt_color[mask, :] - pred_color[mask, :]) ** 2).mean()


def track_depth_loss(gt_depth, pred_depth, pred_var, mask):
    """slams/tracking.py:89-92 (variance-normalised L1)."""
    return (torch.abs(gt_depth - pred_depth) / torch.sqrt(pred_var + 1e-10))[mask].mean()


def track_label_loss(gt_label, pred_logits, mask):
    """slams/tracking.py:94-96."""
    return F.cross_entropy(pred_logits[mask, :], gt_label[mask])
