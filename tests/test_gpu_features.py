"""GPU parity of the 2-D feature branch: feature_matching against the IMPORTED reference's golden
outputs, Decoder.merge against the oracle (values + gradients), and the Mapper's per-frame reference-pose logic."""
import os

import numpy as np
import pytest
import torch

from oracle import feature_ref as fr
from oracle import render_math as rm
from oracle import slam_ref as sr
from util import assert_close, oracle_from_product, randomise_

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_feature_matching_golden(golden_dir):
    from dns_slam_amd.common import feature_matching
    g = np.load(os.path.join(golden_dir, "feature_matching.npz"))
    for ci in range(int(g["n_cases"])):
        p = f"c{ci}_"
        H, W, h, w, Cc, R, P = [int(v) for v in g[p + "dims"]]
        rec = {}

        def merge_fn(refer_p, refer_o, code_pts):
            rec["o"] = refer_o
            return torch.cat((code_pts.mean(0), refer_p.mean(0)), -1)

        out = feature_matching(H, W, _t(g[p + "K"]), _t(g[p + "pts"]).to(DEV), _t(g[p + "w2c"]).to(DEV),
                               _t(g[p + "features"]).to(DEV), merge_fn).cpu()
        ref = _t(g[p + "out"])
        # a projected coordinate within float rounding of x.5 may round to the neighbouring pixel: allow a few rows
        bad = ((out - ref).abs() > 1e-5 * ref.abs().max()).any(-1)
        assert float(bad.float().mean()) <= 0.01, f"case {ci}: {int(bad.sum())} of {P} points differ"
        assert_close(rec["o"].cpu(), _t(g[p + "refer_o"]), rtol=1e-6, what="refer_o")


def test_feature_gather_vs_oracle_large():
    """Replica-like shapes: [3, 64, 120, 160] maps looked up at 240 x 320 resolution, 20 000 points."""
    from dns_slam_amd.common import feature_matching
    g = torch.Generator().manual_seed(0)
    R, C, h, w, H, W, P = 3, 64, 120, 160, 240, 320, 20000
    feats = torch.randn(R, C, h, w, generator=g)
    K = torch.tensor([[200.0, 0, (W - 1) / 2], [0, 200.0, (H - 1) / 2], [0, 0, 1.0]])
    w2c = torch.eye(4).repeat(R, 1, 1)
    w2c[:, :3, 3] = torch.randn(R, 3, generator=g) * 0.2
    pts = torch.randn(P, 3, generator=g) * torch.tensor([1.5, 1.0, 1.0]) + torch.tensor([0.0, 0.0, -3.0])
    ident = lambda p_, o_, c_: c_
    code_o = fr.feature_matching(H, W, K, pts, w2c, feats, ident)
    code_p = feature_matching(H, W, K, pts.to(DEV), w2c.to(DEV), feats.to(DEV), ident).cpu()
    bad = ((code_o - code_p).abs() > 1e-5).any(-1)
    assert float(bad.float().mean()) < 0.002
    assert float((code_o.abs().sum(-1) > 0).float().mean()) > 0.3          # the test does hit valid pixels


def test_feature_gather_four_channels_per_lane_equals_one_lane_per_channel(monkeypatch):
    """The 2-D code lookup runs 16 lanes x float4 per (reference, point) pair when C % 4 == 0 (four pairs per wave; round 5: the
    wave-per-pair kernel issued the projection in all 64 lanes).  Same expressions per channel: bit-identical codes and masks;
    C = 6 (not a multiple of four) still takes the one-lane-per-channel kernel and agrees with the oracle."""
    from dns_slam_amd import ops
    g = torch.Generator().manual_seed(4)
    R, h, w, H, W, P = 3, 60, 80, 120, 160, 5001
    K = torch.tensor([[100.0, 0, (W - 1) / 2], [0, 100.0, (H - 1) / 2], [0, 0, 1.0]])
    w2c = torch.eye(4).repeat(R, 1, 1)
    w2c[:, :3, 3] = torch.randn(R, 3, generator=g) * 0.2
    pts = torch.randn(P, 3, generator=g) * torch.tensor([1.5, 1.0, 1.0]) + torch.tensor([0.0, 0.0, -3.0])
    for C in (64, 128, 8):
        feats = torch.randn(R, h, w, C, generator=g).to(DEV)
        monkeypatch.delenv("DNS_FEATURE_GATHER_LANES", raising=False)
        c4, m4 = ops.feature_gather(pts.to(DEV), w2c.to(DEV), K, feats, H, W)
        monkeypatch.setenv("DNS_FEATURE_GATHER_LANES", "1")
        c1, m1 = ops.feature_gather(pts.to(DEV), w2c.to(DEV), K, feats, H, W)
        assert torch.equal(c4, c1) and torch.equal(m4, m1), f"C = {C}"
        assert float(m4.float().mean()) > 0.2
    monkeypatch.delenv("DNS_FEATURE_GATHER_LANES", raising=False)
    from dns_slam_amd.common import feature_matching
    feats = torch.randn(R, 6, h, w, generator=g)
    ident = lambda p_, o_, c_: c_
    code_o = fr.feature_matching(H, W, K, pts, w2c, feats, ident)
    code_p = feature_matching(H, W, K, pts.to(DEV), w2c.to(DEV), feats.to(DEV), ident).cpu()
    assert float(((code_o - code_p).abs() > 1e-5).any(-1).float().mean()) < 0.002


def test_merge_module_matches_oracle():
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    bound = synthetic.load_bound(synthetic.ROOM0_BOUND)
    cfg = synthetic.default_cfg(hash_size=12, voxel_size=0.2)
    dec = Decoder(cfg["model"], bound, n_class=8).to(DEV)
    randomise_(dec, 2)
    g = torch.Generator().manual_seed(1)
    R, P = 3, 700
    p = (torch.randn(R, P, 3, generator=g) * 2.0)
    o = torch.randn(R, 3, generator=g)
    code = torch.randn(R, P, 64, generator=g)
    gy = torch.randn(P, 32, generator=g)
    pp = p.to(DEV).requires_grad_(True)
    y = dec.merge(pp, o.to(DEV), code.to(DEV))
    (y * gy.to(DEV)).sum().backward()
    params = dec.merge.decoder.params.detach().cpu().clone().requires_grad_(True)
    po = p.clone().requires_grad_(True)
    yo = fr.merge_forward(params, bound, po, o, code)
    (yo * gy).sum().backward()
    assert_close(y.cpu(), yo, what="merge fwd")
    assert_close(dec.merge.decoder.params.grad.cpu()[:112 * 32 + 32 * 32], params.grad[:112 * 32 + 32 * 32], what="merge dparams")
    bad = ((pp.grad.cpu() - po.grad).abs() > 1e-4 * po.grad.abs().max()).any(-1)
    assert float(bad.float().mean()) < 0.002


def test_mapper_feature_branch_matches_oracle():
    """get_target_samples with stem feature maps + refer_frames (slams/mapping.py:533-557) vs the oracle composition."""
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.encoder import ResNet
    from dns_slam_amd.mapping import Mapper
    cam = synthetic.camera(H=60, W=80, fx=60.0, fy=60.0)
    bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=0)
    cfg = synthetic.default_cfg(n_pixels=240, hash_size=14, voxel_size=0.08, smooth_pts=10)
    dec = Decoder(cfg["model"], bound, n_class=8).to(DEV)
    randomise_(dec, 4)
    mapper = Mapper(cfg, dec, bound, cam, device=DEV)
    mapper.set_decoder(frames)
    mapper.is_BA = True
    _, ql, Tl = mapper.set_optimizer(frames)
    frames = dict(frames)
    frames["kf_idx"] = [0, 10, 20, 30]
    # per target frame two references: an older keyframe that is also a target (or a foreign one) and itself (-1)
    refer = {"kf_idx": [[99, -1], [0, -1], [10, -1], [20, -1]],
             "gt_color": torch.stack([torch.stack([frames["gt_color"][max(i - 1, 0)], frames["gt_color"][i]]) for i in range(4)]),
             "est_c2w": torch.stack([torch.stack([frames["est_c2w"][max(i - 1, 0)], frames["est_c2w"][i]]) for i in range(4)])}
    stem = ResNet(seed=3).to(DEV)
    feats = stem(refer["gt_color"].to(DEV))
    assert feats.shape == (4, 2, 64, 30, 40)
    torch.manual_seed(9)
    prep = mapper.prepare_frames(frames)
    pix, jit = mapper.draw_pixels(prep), mapper.draw_jitter()
    s = mapper.get_target_samples(frames, ql, Tl, refer_frames=refer, features=feats, prep=prep, pix_idx=pix, jitter=jit)
    # oracle
    sd = stem.conv_blocks.state_dict()
    fo = fr.stem_forward(refer["gt_color"], sd["conv1.weight"].cpu(), sd["bn1.weight"].cpu(), sd["bn1.bias"].cpu(),
                         sd["bn1.running_mean"].cpu(), sd["bn1.running_var"].cpu())
    assert_close(feats.cpu(), fo, rtol=1e-4, what="stem features")
    K = torch.tensor([[cam["fx"], 0.0, cam["cx"]], [0.0, cam["fy"], cam["cy"]], [0.0, 0.0, 1.0]])
    params = dec.merge.decoder.params.detach().cpu()
    npf = pix.numel() // 4
    camt = (cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]])
    pose = lambda k: torch.cat([torch.cat((rm.rotation_from_quad(ql[k].detach().cpu()), Tl[k].detach().cpu()[:, None]), -1), bottom], 0)
    codes = []
    for i in range(4):
        img5 = torch.cat((frames["gt_color"][i], frames["gt_depth"][i][..., None], frames["gt_label"][i][..., None]), -1)
        fs = sr.frame_samples(img5, ql[i].detach().cpu(), Tl[i].detach().cpu(), camt, bound, pix.cpu()[i * npf:(i + 1) * npf],
                              jit[0][i].cpu(), jit[1][i].cpu(), 32, 15)
        first = refer["est_c2w"][i][0] if i == 0 else pose(i - 1)         # 99 is foreign -> stored pose; else a target's pose
        w2c = torch.stack([torch.inverse(first), torch.inverse(pose(i))])
        merge = lambda p_, o_, c_: fr.merge_forward(params, bound, p_, o_, c_)
        code = fr.feature_matching(cam["H"], cam["W"], K, fs["pts"].flatten(0, 1), w2c, fo[i], merge).reshape(npf, 47, -1)
        codes.append(code * rm.truncation_mask(fs["z_vals"], fs["gt_depth"])[..., None])
    code_o = torch.cat(codes, 0)
    got = s["features"].cpu()
    bad = ((got - code_o).abs() > 1e-4 * code_o.abs().max()).flatten(1).any(-1)
    assert float(bad.float().mean()) < 0.01, f"{int(bad.sum())} rays differ"
    assert float((code_o.abs().sum(-1) > 0).float().mean()) > 0.05


def _stem_setup(layout="per_ray", nn=32, nl=1):
    """A mapper with stem feature maps: 4 target frames, 2 reference views each (a keyframe that is also a target -- or a foreign
    one with a stored pose -- and the frame itself)."""
    from test_gpu_slam import _setup
    from dns_slam_amd.encoder import ResNet
    cfg, bound, cam, frames, dec, mapper = _setup(nn, nl, layout=layout)
    randomise_(dec.merge, 21)
    frames = dict(frames)
    frames["kf_idx"] = [0, 10, 20, 30]
    refer = {"kf_idx": [[99, -1], [0, -1], [10, -1], [20, -1]],
             "gt_color": torch.stack([torch.stack([frames["gt_color"][max(i - 1, 0)], frames["gt_color"][i]]) for i in range(4)]),
             "est_c2w": torch.stack([torch.stack([frames["est_c2w"][max(i - 1, 0)], frames["est_c2w"][i]]) for i in range(4)])}
    stem = ResNet(seed=3).to(DEV)
    feats = stem(refer["gt_color"].to(DEV)).detach()
    return cfg, bound, cam, frames, dec, mapper, refer, feats


def _run_stem(fused, n_iters, nn=32, nl=1):
    from dns_slam_amd.fused_step import MapStep
    cfg, bound, cam, frames, dec, mapper, refer, feats = _stem_setup(nn=nn, nl=nl)
    mapper.static_shapes, mapper.is_BA, mapper.overlap_smooth, mapper.prefetch_draws = True, True, True, True
    opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
    for grp, lr in zip(opt.param_groups, (mapper.lr, mapper.BA_cam_lr, mapper.BA_cam_lr)):
        grp["lr"] = lr
    prep = mapper.prepare_frames(frames)
    torch.manual_seed(77)
    torch.cuda.manual_seed(77)
    hist, grads = [], None
    pool = mapper.fine_decoders.pool
    if fused:
        ms = MapStep(mapper, frames, ql, Tl, prep=prep, features=feats, refer_frames=refer)
        for i in range(n_iters):
            ms.step(last=i == n_iters - 1)
            total, terms = ms.losses()
            hist.append((float(total), {k: float(v) for k, v in terms.items()}))
            if i == 0:
                grads = {"table": ms.g_table, "coarse": ms.g_coarse, "color": ms.g_color, "logit": ms.g_logit, "merge": ms.g_merge,
                         "pool": ms.g_pool.view_as(pool), "quat": ms.g_quat.view(4, 4)[1:], "trans": ms.g_trans.view(4, 3)[1:]}
                grads = {k: v.detach().cpu().clone() for k, v in grads.items()}
        ms.write_back()
    else:
        for i in range(n_iters):
            opt.zero_grad(set_to_none=True)
            s = mapper.get_target_samples(frames, ql, Tl, refer_frames=refer, features=feats, prep=prep)
            loss, terms = mapper.iteration_loss(s, lambda_lt=10.0, smooth=True)
            loss.backward()
            if i == 0:
                grads = {"table": dec.pe_fn.grid_fn.params.grad, "coarse": dec.coarse_fn.decoder.params.grad,
                         "color": dec.out_fn.color_decoder.params.grad, "logit": dec.out_fn.logit_decoder.params.grad,
                         "merge": dec.merge.decoder.params.grad, "pool": pool.grad, "quat": torch.stack([q.grad for q in ql[1:]]),
                         "trans": torch.stack([t.grad for t in Tl[1:]])}
                grads = {k: v.detach().cpu().clone() for k, v in grads.items()}
            opt.step()
            hist.append((float(loss.detach()), {k: float(v) for k, v in terms.items()}))
    torch.cuda.synchronize()
    params = {"table": dec.pe_fn.grid_fn.params, "coarse": dec.coarse_fn.decoder.params, "color": dec.out_fn.color_decoder.params,
              "logit": dec.out_fn.logit_decoder.params, "merge": dec.merge.decoder.params, "pool": pool,
              "quat": torch.stack([q.detach() for q in ql]), "trans": torch.stack([t.detach() for t in Tl])}
    return hist, {k: v.detach().cpu().clone() for k, v in params.items()}, grads, (mapper.lr, mapper.BA_cam_lr)


@pytest.mark.parametrize("net", [(32, 1), (64, 2)])
def test_map_step_with_stem_features_equals_the_autograd_iteration(net):
    """The reference's REAL iteration -- feature_matching + Decoder.merge inside the loop (slams/mapping.py:532-557), Merge's weights
    trained, its OneBlob input carrying pose gradient -- on the fixed launch sequence (MapStep(features=[K, R, C, h, w],
    refer_frames=...)) against the autograd driver on the same draws: the losses of every iteration, the first iteration's
    gradients of every parameter group incl. Merge and the poses, the parameters after the run."""
    n = 5
    ha, pa, ga, lrs = _run_stem(False, n, *net)
    hf, pf, gf, _ = _run_stem(True, n, *net)
    for i, ((la, ta), (lf, tf)) in enumerate(zip(ha, hf)):
        assert abs(la - lf) <= 2e-4 * abs(la), (i, la, lf)
        for k in ta:
            assert abs(ta[k] - tf[k]) <= 2e-4 * max(abs(ta[k]), 1e-6), (i, k, ta[k], tf[k])
    assert float(ga["merge"].abs().max()) > 0 and float(ga["quat"].abs().max()) > 0
    for k in ga:
        assert_close(gf[k], ga[k], rtol=1e-4 if k != "merge" else 2e-4, elementwise=False, what=f"MapStep (stem features) vs autograd: d {k}")
    for k in pa:
        lr = lrs[1] if k in ("quat", "trans") else lrs[0]
        diff = (pf[k] - pa[k]).abs().max().item()
        assert diff <= 1e-4 * pa[k].abs().max().item() + 0.05 * lr * n, (k, diff)


def test_map_step_stem_features_match_the_oracle():
    """MapStep's in-loop 2-D branch against the ORACLE composition (oracle.feature_ref.feature_matching + merge_forward, pinned to
    the imported reference by tests/golden/feature_matching.npz): the truncated code of every sample that reaches the colour /
    logit networks, and the iteration's loss against the oracle's mapping loss on the oracle's code."""
    from dns_slam_amd.fused_step import MapStep
    cfg, bound, cam, frames, dec, mapper, refer, feats = _stem_setup(layout="reference_tiled")
    mapper.static_shapes, mapper.is_BA = True, True
    _, ql, Tl = mapper.set_optimizer(frames)
    prep = mapper.prepare_frames(frames)
    torch.manual_seed(9)
    pix, jit = mapper.draw_pixels(prep), mapper.draw_jitter()
    u_off, u_jit = torch.rand(3), torch.rand((1, 1, 1, 3))
    ms = MapStep(mapper, frames, ql, Tl, prep=prep, features=feats, refer_frames=refer)
    r6 = torch.cat((u_off.reshape(-1), u_jit.reshape(-1))).to(DEV)
    ms.step(draws={"pix": pix, "jitter": jit, "r6": r6})
    torch.cuda.synchronize()
    fo = feats.cpu()
    K = torch.tensor([[cam["fx"], 0.0, cam["cx"]], [0.0, cam["fy"], cam["cy"]], [0.0, 0.0, 1.0]])
    params = dec.merge.decoder.params.detach().cpu()          # (the step has run Adam: use the values it STARTED from)
    npf = pix.numel() // 4
    S = 32 + 15
    camt = (cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    # oracle on the parameters and poses the step started from: rebuild them from a twin set-up (same seeds)
    cfg2, bound2, cam2, frames2, dec2, mapper2, refer2, feats2 = _stem_setup(layout="reference_tiled")
    _, ql2, Tl2 = mapper2.set_optimizer(frames2)
    params = dec2.merge.decoder.params.detach().cpu()
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]])
    pose = lambda k: torch.cat([torch.cat((rm.rotation_from_quad(ql2[k].detach().cpu()), Tl2[k].detach().cpu()[:, None]), -1), bottom], 0)
    frs, codes = [], []
    for i in range(4):
        img5 = torch.cat((frames2["gt_color"][i], frames2["gt_depth"][i][..., None], frames2["gt_label"][i][..., None]), -1)
        fs = sr.frame_samples(img5, ql2[i].detach().cpu(), Tl2[i].detach().cpu(), camt, bound, pix.cpu()[i * npf:(i + 1) * npf],
                              jit[0][i].cpu(), jit[1][i].cpu(), 32, 15)
        first = refer["est_c2w"][i][0] if i == 0 else pose(i - 1)
        w2c = torch.stack([torch.inverse(first), torch.inverse(pose(i))])
        merge = lambda p_, o_, c_: fr.merge_forward(params, bound, p_, o_, c_)
        code = fr.feature_matching(cam["H"], cam["W"], K, fs["pts"].flatten(0, 1), w2c, fo[i], merge).reshape(npf, S, -1)
        code = code * rm.truncation_mask(fs["z_vals"], fs["gt_depth"])[..., None]
        codes.append(code)
        fs = dict(fs)
        fs["features"] = code
        frs.append(fs)
    code_o = torch.cat(codes, 0)
    got = ms.feat[:, ms.hid:].reshape(4 * npf, S, -1).cpu()
    bad = ((got - code_o).abs() > 1e-4 * code_o.abs().max()).flatten(1).any(-1)
    assert float(bad.float().mean()) < 0.01, f"{int(bad.sum())} rays differ"
    assert float((code_o.abs().sum(-1) > 0).float().mean()) > 0.05
    om = oracle_from_product(cfg2, bound2, dec2, mapper2)
    so = sr.mapper_target_samples(frs)
    lo, _, _ = sr.mapping_loss(om, so, sr.LossCfg(smooth_pts=cfg["training"]["smooth_pts"]), u_off, u_jit)
    lm = float(ms.losses()[0])
    assert abs(lm - float(lo)) <= 5e-4 * abs(float(lo)), (lm, float(lo))          # a few samples' code may round to the next pixel


def test_track_step_with_stem_features_equals_the_tracker_loop():
    """TrackStep(features=[1, R, C, h, w], refer_frames={'est_w2c'}) -- feature_matching + the frozen Merge network inside every
    tracking iteration (slams/tracking.py:162-165), its OneBlob input carrying pose gradient -- against Tracker.track_frame's
    autograd loop from the same seed."""
    from test_gpu_slam import _setup
    from dns_slam_amd.encoder import ResNet
    from dns_slam_amd.tracking import Tracker
    cfg, bound, cam, frames, dec, mapper = _setup()
    randomise_(dec.merge, 21)
    tracker = Tracker(cfg, dec, bound, cam, device=DEV)
    tracker.static_shapes = True
    cur = {"gt_color": frames["gt_color"][1], "gt_depth": frames["gt_depth"][1], "gt_label": frames["gt_label"][1]}
    stem = ResNet(seed=3).to(DEV)
    views = torch.stack([frames["gt_color"][0], frames["gt_color"][1], frames["gt_color"][2]])[None].to(DEV)
    feats = stem(views).detach()
    refer = {"est_w2c": torch.stack([torch.inverse(frames["est_c2w"][k].float()) for k in (0, 1, 2)]).to(DEV)}
    est = frames["est_c2w"][1].clone()
    est[:3, 3] += 0.02
    out = []
    for fused in (False, True):
        tracker.use_track_step = fused
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        cam_t, best = tracker.track_frame(cur, est, n_iters=8, features=feats, refer_frames=refer, fused=True)
        out.append((cam_t.detach().cpu().clone(), float(best)))
    (ca, la), (cf, lf) = out
    assert abs(la - lf) <= 1e-4 * abs(la), (la, lf)
    assert float((ca - cf).abs().max()) <= 2e-4, (ca, cf)
