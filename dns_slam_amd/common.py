"""Mirror of the live render-math functions of reference utils/common.py on the gfx950 kernels.

Same names and argument meaning where the reference function is itself a call site of the hot path:
``quad2rotation`` (:406), ``get_rotation_from_quad`` (:447), ``get_camera_from_tensor`` (:432),
``raw2nerf_color`` (:506), ``get_opacity_loss`` (:769).  Ray generation and depth-guided sampling
(``get_samples`` :296, ``sample_along_rays`` :561) are fused into one launch, ``ops.raygen_sample``; the drawn
pixel indices and jitter vectors are explicit inputs there.
"""
import math

import torch

from . import ops


def quad2rotation(quad):
    """utils/common.py:406-429 (w,x,y,z; two_s = 2/|q|^2), allocated on ``quad.device`` (the reference's
    ``.to(quad.get_device())`` fails on CPU tensors).  Tiny [B,4] -> [B,3,3]; kept in torch so
    autograd carries pose gradients for callers outside the fused sampler."""
    qr, qi, qj, qk = quad[:, 0], quad[:, 1], quad[:, 2], quad[:, 3]
    two_s = 2.0 / (quad * quad).sum(-1)
    rows = [1 - two_s * (qj ** 2 + qk ** 2), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
            two_s * (qi * qj + qk * qr), 1 - two_s * (qi ** 2 + qk ** 2), two_s * (qj * qk - qi * qr),
            two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1 - two_s * (qi ** 2 + qj ** 2)]
    return torch.stack(rows, -1).reshape(-1, 3, 3)


def get_rotation_from_quad(quad):
    """utils/common.py:447-458."""
    if quad.dim() == 1:
        return quad2rotation(quad.unsqueeze(0))[0]
    return quad2rotation(quad)


def get_camera_from_tensor(inputs):
    """utils/common.py:432-445: (quat, T) [.,7] -> [.,3,4]."""
    one = inputs.dim() == 1
    if one:
        inputs = inputs.unsqueeze(0)
    quad, T = inputs[:, :4], inputs[:, 4:]
    RT = torch.cat([quad2rotation(quad), T[:, :, None]], 2)
    return RT[0] if one else RT


def get_quad_from_c2w(c2w):
    """Rotation -> unit quaternion (w,x,y,z).  The reference does this on the CPU through
    ``mathutils.Matrix.to_quaternion`` once per ``optimize()`` (utils/common.py:486-504, slams/mapping.py:453);
    same here (host, float64, Shepperd's branch on the largest diagonal term)."""
    R = torch.as_tensor(c2w)[:3, :3].detach().double().cpu()
    m = [[float(R[i, j]) for j in range(3)] for i in range(3)]
    tr = m[0][0] + m[1][1] + m[2][2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        w, x, y, z = 0.25 * s, (m[2][1] - m[1][2]) / s, (m[0][2] - m[2][0]) / s, (m[1][0] - m[0][1]) / s
    elif m[0][0] > m[1][1] and m[0][0] > m[2][2]:
        s = math.sqrt(1.0 + m[0][0] - m[1][1] - m[2][2]) * 2
        w, x, y, z = (m[2][1] - m[1][2]) / s, 0.25 * s, (m[0][1] + m[1][0]) / s, (m[0][2] + m[2][0]) / s
    elif m[1][1] > m[2][2]:
        s = math.sqrt(1.0 + m[1][1] - m[0][0] - m[2][2]) * 2
        w, x, y, z = (m[0][2] - m[2][0]) / s, (m[0][1] + m[1][0]) / s, 0.25 * s, (m[1][2] + m[2][1]) / s
    else:
        s = math.sqrt(1.0 + m[2][2] - m[0][0] - m[1][1]) * 2
        w, x, y, z = (m[1][0] - m[0][1]) / s, (m[0][2] + m[2][0]) / s, (m[1][2] + m[2][1]) / s, 0.25 * s
    return torch.tensor([w, x, y, z], dtype=torch.float32)


def raw2nerf_color(raw, z_vals, rays_d=None, occupancy=True, device=None):
    """utils/common.py:506-537.  Only the occupancy mode exists in the reference's call sites
    (slams/mapping.py:630, slams/tracking.py:209); ``rays_d`` is unused in that mode."""
    if not occupancy:
        raise ValueError("dns_slam_amd.raw2nerf_color: only occupancy=True is on the supported path")
    depth, var, rgb, weights, _ = ops.composite(raw, z_vals, None)
    return depth, var, rgb, weights


def get_opacity_loss(z_vals, depth, occ, truncation=0.2, sigma=0.05):
    """utils/common.py:769-802, branch-free: the reference's ``if count_nonzero(front) > 0 and
    count_nonzero(opacity_mask) > 0`` (a host sync) becomes a 0/1 factor, which gives the same values and the
    same (zero) gradients in the else-branch."""
    bs, n_sample = z_vals.shape
    depth = depth.unsqueeze(-1)
    occ = torch.sigmoid(10 * occ).reshape(bs, n_sample)
    front = (z_vals < (depth - truncation)).to(z_vals.dtype)
    back = (z_vals > (depth + truncation)).to(z_vals.dtype)
    dmask = (depth > 0.0).to(z_vals.dtype)
    omask = (1.0 - front) * (1.0 - back) * dmask
    flag = ((front.sum() > 0) & (omask.sum() > 0)).to(z_vals.dtype)
    fs_loss = flag * ((occ * front * dmask) ** 2).mean()
    pseudo = 0.5 * torch.exp(-0.5 * ((z_vals - depth) / sigma) ** 2)
    opacity_loss = flag * ((occ * omask - pseudo * omask) ** 2).mean()
    return fs_loss, opacity_loss


def sample_along_rays(gt_depth, n_samples, n_surface, far_bb, device=None, jitter=None):
    """utils/common.py:561-599, same signature; the two ``torch.rand(n_surface)`` draws (and the forced 0.5) come
    from the CPU generator in the reference's order unless ``jitter=(t_surf, t_zero)`` is given."""
    dev = gt_depth.device
    if jitter is None:
        t = torch.rand(n_surface)
        if not torch.any(t == 0.5):
            t[n_surface // 2 + 1] = 0.5
        jitter = (t, torch.rand(n_surface))
    tu = torch.linspace(0.0, 1.0, steps=n_samples, device=dev) if n_samples > 0 else None
    return ops.sample_along_rays(gt_depth, far_bb, tu, jitter[0].to(dev), jitter[1].to(dev))


_NHWC_CACHE = {}


def features_channels_last(features):
    """[R,C,h,w] stem features -> contiguous [R,h,w,C], cached per tensor (the reference up-samples the SAME maps in
    every feature_matching call of an optimize(); the re-layout is done once)."""
    key = (features.data_ptr(), features._version, tuple(features.shape))
    hit = _NHWC_CACHE.get("k")
    if hit is not None and hit[0] == key:
        return hit[1]
    nhwc = features.detach().permute(0, 2, 3, 1).contiguous().float()
    _NHWC_CACHE["k"] = (key, nhwc)
    return nhwc


def feature_matching(H, W, K, pts_, refer_w2c, features, merge_fn):
    """utils/common.py:645-679, same signature: project ``pts_`` [P,3] into the R reference frames, look the code up in the
    [R,C,h,w] stem feature maps (bilinear-at-rounded-pixel, fused in csrc/feature.hip instead of F.interpolate to full
    resolution), zero it where the projection is invalid, and merge the R views with ``merge_fn(pts - o, o, code)``."""
    code, _mask = ops.feature_gather(pts_, refer_w2c, K, features_channels_last(features), H, W)
    refer_c2w = torch.inverse(refer_w2c)
    refer_o = refer_c2w[:, :3, 3]
    refer_p = pts_[None, :, :] - refer_o[:, None, :]
    return merge_fn(refer_p, refer_o, code)
