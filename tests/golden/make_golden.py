"""Generate the golden vectors under tests/golden/ from the IMPORTED reference.

Run once, in the build container only (the reference never travels):

    cd /root/reference && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

Recipe = SURVEY.md Appendix D: ``utils/common.py`` imports ``mathutils`` at module level
(common.py:12) only for two pose-conversion helpers this script never calls, so an empty
in-memory module object is registered under that name before the import; nothing is
written next to the reference sources.  Outputs are plain arrays (inputs + expected
outputs); no reference source text is stored.

Functions exercised (all from reference utils/common.py): get_samples (:296),
get_samples_by_class (:353), get_samples_by_uniq_class (:364), get_all_rays (:540), sample_along_rays (:561),
raw2nerf_color (:506), get_opacity_loss (:769).
"""
import os
import sys
import types

import numpy as np
import torch

OUT = os.path.dirname(os.path.abspath(__file__))

_m = types.ModuleType("mathutils")
_m.Matrix = object
sys.modules["mathutils"] = _m
sys.path.insert(0, "/root/reference")
import utils.common as C  # noqa: E402

import warnings  # noqa: E402
warnings.filterwarnings("ignore")


def rand_pose(seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(4, generator=g)
    q = q / q.norm()
    r, i, j, k = q.tolist()
    R = torch.tensor([[1 - 2 * (j * j + k * k), 2 * (i * j - k * r), 2 * (i * k + j * r)],
                      [2 * (i * j + k * r), 1 - 2 * (i * i + k * k), 2 * (j * k - i * r)],
                      [2 * (i * k - j * r), 2 * (j * k + i * r), 1 - 2 * (i * i + j * j)]], dtype=torch.float32)
    T = torch.randn(3, generator=g)
    return R, T


def make_image(H, W, seed, n_class=6):
    g = torch.Generator().manual_seed(seed)
    color = torch.rand(H, W, 3, generator=g)
    depth = torch.rand(H, W, generator=g) * 4 + 0.3
    depth[torch.rand(H, W, generator=g) < 0.05] = 0.0
    label = torch.randint(0, n_class, (H, W), generator=g).float()
    label[0, 0] = float(n_class)          # a class with exactly one pixel (common.py:324)
    return torch.cat((color, depth[..., None], label[..., None]), -1)


def golden_get_samples():
    out = {}
    cases = [(48, 64, 0, 48, 0, 64, 256, 0), (12, 16, 0, 12, 0, 16, 40, 1),
             (60, 80, 20, 40, 20, 60, 128, 2)]   # the last = tracker-style 20-px border window (tracking.py:137)
    for ci, (H, W, H0, H1, W0, W1, n, seed) in enumerate(cases):
        img = make_image(H, W, 10 + ci)
        R, T = (torch.eye(3), torch.zeros(3)) if ci == 0 else rand_pose(20 + ci)
        fx, fy, cx, cy = 500.0, 510.0, (W - 1) / 2.0, (H - 1) / 2.0
        torch.manual_seed(seed)
        idx = torch.randint((H1 - H0) * (W1 - W0), (n,))
        torch.manual_seed(seed)
        ro, rd, smp = C.get_samples(H0, H1, W0, W1, n, H, W, fx, fy, cx, cy, R, T, img, "cpu")
        p = f"c{ci}_"
        out.update({p + "image": img.numpy(), p + "R": R.numpy(), p + "T": T.numpy(),
                    p + "cam": np.array([H, W, fx, fy, cx, cy]), p + "window": np.array([H0, H1, W0, W1]),
                    p + "indices": idx.numpy(), p + "rays_o": ro.numpy(), p + "rays_d": rd.numpy(),
                    p + "sample": smp.numpy()})
    np.savez_compressed(os.path.join(OUT, "get_samples.npz"), **out)


def golden_by_class():
    out = {}
    for ci, (H, W, n, seed) in enumerate([(24, 32, 50, 3), (12, 16, 33, 4)]):
        img = make_image(H, W, 30 + ci)
        R, T = rand_pose(40 + ci)
        fx, fy, cx, cy = 50.0, 50.0, (W - 1) / 2.0, (H - 1) / 2.0
        torch.manual_seed(seed)
        ro, rd, smp = C.get_samples_by_class(0, H, 0, W, n, H, W, fx, fy, cx, cy, R, T, img, "cpu")
        p = f"c{ci}_"
        out.update({p + "image": img.numpy(), p + "R": R.numpy(), p + "T": T.numpy(),
                    p + "cam": np.array([H, W, fx, fy, cx, cy]), p + "n": np.array(n), p + "seed": np.array(seed),
                    p + "rays_o": ro.numpy(), p + "rays_d": rd.numpy(), p + "sample": smp.numpy()})
    np.savez_compressed(os.path.join(OUT, "get_samples_by_class.npz"), **out)


def golden_class_picks():
    """Pixel INDICES drawn by get_samples_by_class (:353, select_by_class :307-338) and get_samples_by_uniq_class (:364-403)
    under a seeded CPU generator: with R = I, T = 0, fx = fy = 1, cx = cy = 0 the returned rays_d are (i, -j, -1), so the
    drawn (column, row) pairs -- hence the flat indices -- are read off the reference's own output."""
    out = {}
    ci = 0
    R, T = torch.eye(3), torch.zeros(3)
    for (H, W, n, seed, class_dict) in [(24, 32, 50, 3, None), (12, 16, 33, 4, None), (24, 32, 50, 5, [2.0, 6.0, 0.0, 9.0]),
                                        (12, 16, 31, 6, [6.0, 1.0]), (20, 20, 64, 7, [0.0, 1.0, 2.0, 3.0, 4.0, 5.0])]:
        img = make_image(H, W, 60 + ci)          # label 6 has exactly one pixel (0, 0); label 9 is absent
        torch.manual_seed(seed)
        if class_dict is None:
            ro, rd, smp = C.get_samples_by_class(0, H, 0, W, n, H, W, 1.0, 1.0, 0.0, 0.0, R, T, img, "cpu")
        else:
            ro, rd, smp = C.get_samples_by_uniq_class(0, H, 0, W, n, H, W, 1.0, 1.0, 0.0, 0.0, R, T, img, class_dict, "cpu")
        idx = (-rd[:, 1]).round().long() * W + rd[:, 0].round().long()
        assert torch.equal(img.reshape(-1, 5)[idx], smp)
        p = f"c{ci}_"
        out.update({p + "image": img.numpy(), p + "n": np.array(n), p + "seed": np.array(seed),
                    p + "class_dict": np.array(class_dict if class_dict is not None else [], dtype=np.float64),
                    p + "uniq": np.array(0 if class_dict is None else 1), p + "indices": idx.numpy()})
        ci += 1
    out["n_cases"] = np.array(ci)
    np.savez_compressed(os.path.join(OUT, "class_picks.npz"), **out)


def golden_all_rays():
    H, W = 6, 8
    R, T = rand_pose(50)
    c2w = torch.eye(4)
    c2w[:3, :3] = R
    c2w[:3, 3] = T
    ro, rd = C.get_all_rays(H, W, 7.0, 7.5, 3.5, 2.5, c2w, "cpu")
    np.savez_compressed(os.path.join(OUT, "get_all_rays.npz"), c2w=c2w.numpy(), cam=np.array([H, W, 7.0, 7.5, 3.5, 2.5]),
                        rays_o=ro.numpy(), rays_d=rd.numpy())


def golden_sample_along_rays():
    out = {}
    ci = 0
    for seed in (0, 1, 2):
        for (ns, nf) in ((32, 15), (48, 16), (22, 10), (96, 32), (0, 15)):
            g = torch.Generator().manual_seed(100 + seed)
            n = 37
            d = torch.rand(n, generator=g) * 5 + 0.2
            d[torch.rand(n, generator=g) < 0.15] = 0.0
            if seed == 2:
                d[:] = torch.where(torch.arange(n) % 2 == 0, d, torch.zeros_like(d))
            far = torch.rand(n, 1, generator=g, dtype=torch.float64) * 8 - 0.5     # some negative -> clamp to 0
            far[3, 0] = 1e3                                                          # clamp to 1.2*max
            torch.manual_seed(seed)
            t = torch.rand(nf)
            t0 = torch.rand(nf)
            torch.manual_seed(seed)
            z = C.sample_along_rays(d, ns, nf, far.clone(), "cpu")
            p = f"c{ci}_"
            out.update({p + "depth": d.numpy(), p + "far_bb": far.numpy(), p + "n": np.array([ns, nf]),
                        p + "t_raw": t.numpy(), p + "t_zero": t0.numpy(), p + "z": z.numpy()})
            ci += 1
    out["n_cases"] = np.array(ci)
    np.savez_compressed(os.path.join(OUT, "sample_along_rays.npz"), **out)


def golden_raw2nerf():
    out = {}
    for ci, (N, S, seed) in enumerate([(9, 47, 0), (5, 64, 1), (3, 1, 2), (4, 128, 3)]):
        g = torch.Generator().manual_seed(200 + seed)
        raw = torch.randn(N, S, 4, generator=g)
        raw[..., 3] *= 0.3
        raw[..., :3] = torch.sigmoid(raw[..., :3])
        z = torch.sort(torch.rand(N, S, generator=g) * 4 + 0.1, -1)[0]
        rays_d = torch.randn(N, 3, generator=g)
        raw.requires_grad_(True)
        depth, var, rgb, w = C.raw2nerf_color(raw, z, rays_d, device="cpu")
        gd, gv, gr = torch.randn(N, generator=g), torch.randn(N, generator=g), torch.randn(N, 3, generator=g)
        gw = torch.randn(N, S, generator=g)
        (depth * gd).sum().add((var * gv).sum()).add((rgb * gr).sum()).add((w * gw).sum()).backward()
        p = f"c{ci}_"
        out.update({p + "raw": raw.detach().numpy(), p + "z": z.numpy(), p + "depth": depth.detach().numpy(),
                    p + "var": var.detach().numpy(), p + "rgb": rgb.detach().numpy(), p + "weights": w.detach().numpy(),
                    p + "g_depth": gd.numpy(), p + "g_var": gv.numpy(), p + "g_rgb": gr.numpy(), p + "g_w": gw.numpy(),
                    p + "grad_raw": raw.grad.numpy()})
    out["n_cases"] = np.array(4)
    np.savez_compressed(os.path.join(OUT, "raw2nerf_color.npz"), **out)


def golden_opacity():
    out = {}
    g = torch.Generator().manual_seed(300)
    N, S = 11, 47
    z = torch.sort(torch.rand(N, S, generator=g) * 4 + 0.1, -1)[0]
    depth = torch.rand(N, generator=g) * 3 + 0.5
    depth[2] = 0.0
    occ = torch.randn(N * S, generator=g) * 0.2
    for ci, (dd, trunc) in enumerate([(depth, 0.05), (depth, 0.2), (torch.zeros(N), 0.05)]):
        o = occ.clone().requires_grad_(True)
        fs, op = C.get_opacity_loss(z, dd, o, trunc)
        p = f"c{ci}_"
        if fs.requires_grad:
            (3.0 * fs + 7.0 * op).backward()
            out[p + "grad_occ"] = o.grad.numpy()
        out.update({p + "z": z.numpy(), p + "depth": dd.numpy(), p + "occ": occ.numpy(), p + "trunc": np.array(trunc),
                    p + "fs": np.array(float(fs)), p + "op": np.array(float(op))})
    out["n_cases"] = np.array(3)
    np.savez_compressed(os.path.join(OUT, "get_opacity_loss.npz"), **out)


def golden_feature_matching():
    """feature_matching (utils/common.py:645-679) with a recording merge_fn: returns [mean_r(code) | mean_r(refer_p)]."""
    out = {}
    for ci, (H, W, h, w, Cc, R, P, seed) in enumerate([(12, 16, 6, 8, 4, 2, 300, 0), (24, 32, 12, 16, 8, 3, 500, 1)]):
        g = torch.Generator().manual_seed(400 + seed)
        feats = torch.randn(R, Cc, h, w, generator=g)
        fx = fy = float(W)
        K = torch.tensor([[fx, 0.0, (W - 1) / 2.0], [0.0, fy, (H - 1) / 2.0], [0.0, 0.0, 1.0]])
        w2c = []
        for r in range(R):
            Rm, T = rand_pose(410 + 10 * seed + r)
            c2w = torch.eye(4)
            c2w[:3, :3] = Rm
            c2w[:3, 3] = T * 0.3
            w2c.append(torch.inverse(c2w))
        w2c = torch.stack(w2c, 0)
        pts = torch.randn(P, 3, generator=g) * 2.0
        rec = {}

        def merge_fn(refer_p, refer_o, code_pts):
            rec["refer_o"] = refer_o.clone()
            return torch.cat((code_pts.mean(0), refer_p.mean(0)), -1)

        res = C.feature_matching(H, W, K, pts, w2c, feats, merge_fn)
        p = f"c{ci}_"
        out.update({p + "dims": np.array([H, W, h, w, Cc, R, P]), p + "K": K.numpy(), p + "w2c": w2c.numpy(),
                    p + "features": feats.numpy(), p + "pts": pts.numpy(), p + "out": res.numpy(),
                    p + "refer_o": rec["refer_o"].numpy()})
    out["n_cases"] = np.array(2)
    np.savez_compressed(os.path.join(OUT, "feature_matching.npz"), **out)


if __name__ == "__main__":
    golden_feature_matching()
    golden_get_samples()
    golden_by_class()
    golden_class_picks()
    golden_all_rays()
    golden_sample_along_rays()
    golden_raw2nerf()
    golden_opacity()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
