"""Pin the oracle (oracle/render_math.py) to the golden vectors produced by the IMPORTED
reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import torch

from oracle import render_math as rm


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_get_samples_bit_exact(golden_dir):
    g = _load(golden_dir, "get_samples.npz")
    for ci in range(3):
        p = f"c{ci}_"
        H, W, fx, fy, cx, cy = g[p + "cam"].tolist()
        H0, H1, W0, W1 = [int(v) for v in g[p + "window"]]
        idx = _t(g[p + "indices"])
        img = _t(g[p + "image"])
        i, j = rm.uv_from_indices(idx, H0, H1, W0, W1)
        px = rm.gather_pixels(idx, img, H0, H1, W0, W1)
        ro, rd = rm.rays_from_uv(i, j, _t(g[p + "R"]), _t(g[p + "T"]), fx, fy, cx, cy)
        assert torch.equal(px, _t(g[p + "sample"]))
        assert torch.equal(ro, _t(g[p + "rays_o"]))
        assert torch.equal(rd, _t(g[p + "rays_d"]))


def test_get_samples_by_class_replay(golden_dir):
    g = _load(golden_dir, "get_samples_by_class.npz")
    for ci in range(2):
        p = f"c{ci}_"
        H, W, fx, fy, cx, cy = g[p + "cam"].tolist()
        H, W = int(H), int(W)
        img = _t(g[p + "image"])
        torch.manual_seed(int(g[p + "seed"]))
        idx = rm.class_balanced_indices(img[..., -1], int(g[p + "n"]))
        i, j = rm.uv_from_indices(idx, 0, H, 0, W)
        px = rm.gather_pixels(idx, img, 0, H, 0, W)
        ro, rd = rm.rays_from_uv(i, j, _t(g[p + "R"]), _t(g[p + "T"]), fx, fy, cx, cy)
        assert torch.equal(px, _t(g[p + "sample"]))
        assert torch.equal(rd, _t(g[p + "rays_d"]))
        assert torch.equal(ro, _t(g[p + "rays_o"]))


def test_get_all_rays(golden_dir):
    g = _load(golden_dir, "get_all_rays.npz")
    H, W, fx, fy, cx, cy = g["cam"].tolist()
    H, W = int(H), int(W)
    c2w = _t(g["c2w"])
    idx = torch.arange(H * W)
    i, j = rm.uv_from_indices(idx, 0, H, 0, W)
    ro, rd = rm.rays_from_uv(i, j, c2w[:3, :3], c2w[:3, 3], fx, fy, cx, cy)
    assert torch.equal(rd.reshape(H, W, 3), _t(g["rays_d"]))
    assert torch.equal(ro.reshape(H, W, 3), _t(g["rays_o"]))


def test_sample_along_rays_bit_exact(golden_dir):
    g = _load(golden_dir, "sample_along_rays.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        ns, nf = [int(v) for v in g[p + "n"]]
        t = _t(g[p + "t_raw"]).clone()
        if not torch.any(t == 0.5):
            t[nf // 2 + 1] = 0.5
        z = rm.sample_along_rays(_t(g[p + "depth"]), ns, nf, _t(g[p + "far_bb"]), t, _t(g[p + "t_zero"]))
        assert z.dtype == torch.float32
        assert torch.equal(z, _t(g[p + "z"])), f"case {ci}"


def test_sample_along_rays_replay_draw_order(golden_dir):
    g = _load(golden_dir, "sample_along_rays.npz")
    # seeds cycle 0,1,2 with 5 shape cases each (make_golden.py)
    for ci in (0, 6, 12):
        p = f"c{ci}_"
        ns, nf = [int(v) for v in g[p + "n"]]
        torch.manual_seed(ci // 5)
        t, t0 = rm.surface_jitter_replay(nf)
        z = rm.sample_along_rays(_t(g[p + "depth"]), ns, nf, _t(g[p + "far_bb"]), t, t0)
        assert torch.equal(z, _t(g[p + "z"]))


def test_raw2nerf_color_and_grad(golden_dir):
    g = _load(golden_dir, "raw2nerf_color.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        raw = _t(g[p + "raw"]).clone().requires_grad_(True)
        z = _t(g[p + "z"])
        depth, var, rgb, w = rm.raw2nerf_color(raw, z)
        for a, k in ((depth, "depth"), (var, "var"), (rgb, "rgb"), (w, "weights")):
            assert torch.equal(a.detach(), _t(g[p + k])), (ci, k)
        loss = (depth * _t(g[p + "g_depth"])).sum() + (var * _t(g[p + "g_var"])).sum() \
            + (rgb * _t(g[p + "g_rgb"])).sum() + (w * _t(g[p + "g_w"])).sum()
        loss.backward()
        torch.testing.assert_close(raw.grad, _t(g[p + "grad_raw"]), rtol=1e-6, atol=1e-7)


def test_opacity_loss(golden_dir):
    g = _load(golden_dir, "get_opacity_loss.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        occ = _t(g[p + "occ"]).clone().requires_grad_(True)
        fs, op = rm.opacity_loss(_t(g[p + "z"]), _t(g[p + "depth"]), occ, float(g[p + "trunc"]))
        assert abs(float(fs) - float(g[p + "fs"])) <= 1e-7 * max(1.0, abs(float(g[p + "fs"])))
        assert abs(float(op) - float(g[p + "op"])) <= 1e-7 * max(1.0, abs(float(g[p + "op"])))
        if p + "grad_occ" in g:
            (3.0 * fs + 7.0 * op).backward()
            torch.testing.assert_close(occ.grad, _t(g[p + "grad_occ"]), rtol=1e-6, atol=1e-9)
        else:
            assert not fs.requires_grad      # the zero-branch returns constants (common.py:799-800)


def test_feature_matching_golden(golden_dir):
    """2-D feature branch (utils/common.py:645-679) against the imported reference with a recording merge_fn."""
    from oracle import feature_ref as fr
    g = _load(golden_dir, "feature_matching.npz")
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        H, W, h, w, Cc, R, P = [int(v) for v in g[p + "dims"]]
        rec = {}

        def merge_fn(refer_p, refer_o, code_pts):
            rec["o"] = refer_o
            return torch.cat((code_pts.mean(0), refer_p.mean(0)), -1)

        out = fr.feature_matching(H, W, _t(g[p + "K"]), _t(g[p + "pts"]), _t(g[p + "w2c"]), _t(g[p + "features"]), merge_fn)
        assert torch.equal(out, _t(g[p + "out"]))
        assert torch.equal(rec["o"], _t(g[p + "refer_o"]))
