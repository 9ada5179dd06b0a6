"""Mirror of the live render-math functions of reference utils/common.py on the gfx950 kernels.

Same names and argument meaning where the reference function is itself a call site of the hot path:
``quad2rotation`` (:406), ``get_rotation_from_quad`` (:447), ``get_camera_from_tensor`` (:432),
``raw2nerf_color`` (:506), ``get_opacity_loss`` (:769), ``get_samples`` (:296), ``get_samples_by_class`` (:353),
``get_samples_by_uniq_class`` (:364), ``get_all_rays`` (:540), ``sample_along_rays`` (:561), ``feature_matching`` (:645).
Inside the optimise steps ray generation and depth-guided sampling are fused into one launch, ``ops.raygen_sample``; the
drawn pixel indices and jitter vectors are explicit inputs there.
"""
import math

import torch

from . import ops


# ----------------------------------------------------------------------------- pixel sampling + rays (a1 - a3)
def _class_pick(label_flat, n, device, class_list=None):
    """The index draws of select_by_class (utils/common.py:313-328) / get_samples_by_uniq_class (:378-394), in the
    reference's order and with the reference's shapes -- per class, ascending: ``torch.randint(k, (m,), device=device)``
    over the class's k pixels (ascending pixel order, as ``nonzero``), the first class takes the remainder
    ``n - (n // n_class) * (n_class - 1)``, a class with exactly ONE pixel is repeated without a draw, and (uniq variant) a
    class of ``class_list`` that is absent from the image is skipped -- so the same generator state yields the same indices.
    One host sync (the per-class pixel counts), where the reference syncs once per class."""
    order = torch.argsort(label_flat, stable=True)                       # pixels grouped by class, ascending inside a class
    classes, cnt = torch.unique_consecutive(label_flat[order], return_counts=True)
    cls_host, cnt_host = classes.tolist(), cnt.tolist()
    start_of = {}
    o = 0
    for c, k in zip(cls_host, cnt_host):
        start_of[float(c)] = (o, k)
        o += k
    wanted = cls_host if class_list is None else [float(c) for c in class_list]
    n_class = len(wanted)
    n_k = n // n_class
    parts = []
    for i, c in enumerate(wanted):
        m = n - n_k * (n_class - 1) if i == 0 else n_k
        st, k = start_of.get(float(c), (0, 0))
        if k == 1:
            parts.append(order[st:st + 1].repeat(m))
        elif k > 1:
            parts.append(order[st + torch.randint(k, (m,), device=device)])
    return torch.cat(parts, dim=-1)


def _rays_for(indices, H0, H1, W0, W1, H, W, fx, fy, cx, cy, R, T, color, device):
    R = torch.as_tensor(R, device=device)
    T = torch.as_tensor(T, device=device)
    img = color.to(device)
    rays_o, rays_d, sample, _ = ops.rays_from_pixels(R, T, indices, img, (fx, fy, cx, cy), (img.shape[0], img.shape[1]),
                                                     (H0, H1, W0, W1))
    return rays_o, rays_d, sample


def get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, R, T, color, device):
    """utils/common.py:296-304, same signature: n uniformly drawn pixels of the window [H0,H1)x[W0,W1) of ``color``
    [H,W,C] (rgb | depth | label) -> (rays_o [n,3], rays_d [n,3], sample [n,C]).  The draw is the reference's
    (``torch.randint(HW_window, (n,), device=device)``, :274); gather and rays are one HIP launch."""
    indices = torch.randint((H1 - H0) * (W1 - W0), (n,), device=device)
    return _rays_for(indices, H0, H1, W0, W1, H, W, fx, fy, cx, cy, R, T, color, device)


def get_samples_by_class(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, R, T, color, device):
    """utils/common.py:353-361 (select_by_class :307-338), same signature: n pixels drawn class-balanced over the labels
    present in the window (last channel of ``color``)."""
    label = color[H0:H1, W0:W1, -1].to(device).reshape(-1)
    indices = _class_pick(label, n, device)
    return _rays_for(indices, H0, H1, W0, W1, H, W, fx, fy, cx, cy, R, T, color, device)


def get_samples_by_uniq_class(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, R, T, color, class_dict, device):
    """utils/common.py:364-403, same signature: class-balanced over the classes of ``class_dict`` only (decoder warm-up,
    slams/mapping.py:787); a class absent from the window is skipped (its quota is NOT redistributed, as in the reference)."""
    label = color[H0:H1, W0:W1, -1].to(device).reshape(-1)
    indices = _class_pick(label, n, device, class_list=list(class_dict))
    return _rays_for(indices, H0, H1, W0, W1, H, W, fx, fy, cx, cy, R, T, color, device)


def get_all_rays(H, W, fx, fy, cx, cy, c2w, device):
    """utils/common.py:540-559, same signature: (rays_o, rays_d) [H,W,3] of every pixel."""
    c2w = torch.as_tensor(c2w).to(device).float()
    rays_o, rays_d, _, _ = ops.rays_from_pixels(c2w[:3, :3], c2w[:3, 3], None, None, (fx, fy, cx, cy), (H, W), (0, H, 0, W))
    return rays_o.reshape(H, W, 3), rays_d.reshape(H, W, 3)


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
