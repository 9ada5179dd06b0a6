"""Oracle: restatement of the 2-D feature branch.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  ``feature_matching`` / ``feature_searching`` follow reference
``utils/common.py:632-679`` and are PINNED by ``tests/golden/feature_matching.npz`` (outputs of the imported reference with
a recording ``merge_fn``).  ``merge_forward`` follows ``Merge.forward`` (``models/decoder.py:67-77``) on top of the
(parity-unpinned) OneBlob / MLP restatement of ``tcnn_ref``; ``stem_forward`` is the frozen ResNet-18 stem actually
executed by ``models/layers.py:95-98`` (conv 7x7 stride 2 pad 3, batch-norm in eval mode, ReLU).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import tcnn_ref as tr


def feature_searching(pts, features, H, W):
    """utils/common.py:632-642: per reference frame, features[:, h, w] at clamped integer pixels."""
    out = []
    for i in range(features.shape[0]):
        h = pts[i, :, 1].clamp(0, H - 1)
        w = pts[i, :, 0].clamp(0, W - 1)
        out.append(features[i][:, h, w])
    return torch.stack(out, dim=0)


def project(H, W, K, pts_, refer_w2c):
    """utils/common.py:647-663: rounded pixel coordinates [R,P,2] (int64, zeroed where invalid) and the validity mask."""
    ones = torch.ones(pts_.shape[0], 1)
    pts = torch.cat((pts_, ones), dim=-1)
    v = torch.matmul(refer_w2c, pts.permute(1, 0))
    v[:, 1, :] *= -1
    v[:, 2, :] *= -1
    proj_depth = v[:, 2, :]
    v = torch.matmul(K[None, :, :], v[:, :3, :])
    uv = v[:, :2, :] / (v[:, 2:3, :] + 1e-5)
    uv = torch.round(uv.permute(0, 2, 1))
    mask = (uv[:, :, 0] > 0) * (uv[:, :, 0] < W - 1) * (uv[:, :, 1] > 0) * (uv[:, :, 1] < H - 1) * (proj_depth > 0)
    uv = uv * mask[:, :, None]
    return uv.to(torch.int64), mask


def feature_matching(H, W, K, pts_, refer_w2c, features, merge_fn):
    """utils/common.py:645-679: bilinear upsample (align_corners) to full resolution, nearest rounded pixel of each
    projected point, zero code where the projection is invalid, then ``merge_fn(pts - refer_o, refer_o, code)``."""
    features = F.interpolate(features, size=[H, W], mode="bilinear", align_corners=True)
    uv, mask = project(H, W, K, pts_, refer_w2c)
    code = feature_searching(uv, features, H, W).permute(0, 2, 1)
    refer_c2w = torch.inverse(refer_w2c)
    refer_o = refer_c2w[:, :3, 3]
    refer_p = pts_[None, :, :] - refer_o[:, None, :]
    code = code * mask[:, :, None]
    return merge_fn(refer_p, refer_o, code)


def merge_forward(params, bound, p, o, features, n_bins=16, n_neurons=32, n_hidden_layers=1, hidden_dim=32):
    """models/decoder.py:67-77 (NB the RELATIVE vector p is normalised with the scene bound as if absolute, D14)."""
    n_refer, n_points, Cc = features.shape
    p = (p - bound[:, 0]) / (bound[:, 1] - bound[:, 0])
    pe = tr.oneblob_forward(p.flatten(0, 1).float(), n_bins)
    x = torch.cat((pe, features.flatten(0, 1)), -1)
    lat = tr.mlp_forward(x, params, 3 * n_bins + Cc, hidden_dim, n_neurons, n_hidden_layers)
    return torch.mean(lat.reshape(n_refer, n_points, -1), 0)


def stem_forward(images, conv_w, bn_w, bn_b, bn_mean, bn_var, eps=1e-5):
    """models/encoder.py:9-17 + models/layers.py:95-98: images [B,N,H,W,3] -> [B,N,64,H/2,W/2]."""
    B, N = images.shape[:2]
    x = images.flatten(0, 1).permute(0, 3, 1, 2)
    x = F.conv2d(x, conv_w, None, stride=2, padding=3)
    x = F.batch_norm(x, bn_mean, bn_var, bn_w, bn_b, False, 0.0, eps)
    x = F.relu(x)
    return x.reshape(B, N, *x.shape[1:])
