"""GPU parity at the reference's interface level: Mapper.get_target_samples / renderer / the seven-term loss and its
gradients (grid, every MLP, per-class fine decoders, quaternion, translation), Tracker.renderer / losses, against the
oracle on identical parameters, indices and jitter.  Tolerance 1e-4 relative (BASELINE.json)."""
import pytest
import torch

from oracle import render_math as rm
from oracle import slam_ref as sr
from util import assert_close, assert_pose_grad_close, oracle_from_product, randomise_, rel_err, table_level_groups

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(n_neurons=32, n_hidden_layers=1, n_pixels=360, ns_ray=32, nsurf=15, layout="reference_tiled", seed=0):
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.mapping import Mapper
    cam = synthetic.camera(H=60, W=80, fx=60.0, fy=60.0)
    bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=seed)
    cfg = synthetic.default_cfg(n_pixels=n_pixels, n_samples_ray=ns_ray, n_surface_ray=nsurf, n_frames=4, hash_size=14,
                                voxel_size=0.08, n_neurons=n_neurons, n_hidden_layers=n_hidden_layers, smooth_pts=12)
    dec = Decoder(cfg["model"], bound, n_class=8).to(DEV)
    mapper = Mapper(cfg, dec, bound, cam, device=DEV, label_layout=layout)
    mapper.set_decoder(frames)
    randomise_(dec, 11, scale=1.0)
    with torch.no_grad():
        dec.pe_fn.grid_fn.params.mul_(2000.0)            # U(-1e-4,1e-4) init would hide the grid in rounding noise
    randomise_([mapper.fine_decoders.pool], 12)
    return cfg, bound, cam, frames, dec, mapper


def _oracle_samples(frames, quats, Ts, cam, bound, pix_idx, jitter, npf, ns_ray, nsurf):
    camt = (cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    out = []
    for f in range(4):
        img5 = torch.cat((frames["gt_color"][f], frames["gt_depth"][f][..., None], frames["gt_label"][f][..., None]), -1)
        j0, j1 = (jitter[0][f], jitter[1][f]) if jitter[0].dim() == 2 else jitter     # one pair per frame (reference) or shared
        out.append(sr.frame_samples(img5, quats[f], Ts[f], camt, bound, pix_idx[f * npf:(f + 1) * npf], j0, j1, ns_ray, nsurf))
    return sr.mapper_target_samples(out)


def test_get_target_samples_matches_oracle():
    cfg, bound, cam, frames, dec, mapper = _setup()
    mapper.is_BA = True
    _, quad_list, T_list = mapper.set_optimizer(frames)
    prep = mapper.prepare_frames(frames)
    torch.manual_seed(3)
    pix = mapper.draw_pixels(prep)
    jit = mapper.draw_jitter()
    s = mapper.get_target_samples(frames, quad_list, T_list, prep=prep, pix_idx=pix, jitter=jit)
    npf = pix.numel() // 4
    so = _oracle_samples(frames, [q.detach().cpu() for q in quad_list], [t.detach().cpu() for t in T_list], cam, bound,
                         pix.cpu(), (jit[0].cpu(), jit[1].cpu()), npf, 32, 15)
    assert torch.equal(s["gt_label"].cpu(), so["gt_label"])
    assert torch.equal(s["gt_depth"].cpu(), so["gt_depth"]) and torch.equal(s["gt_color"].cpu(), so["gt_color"])
    assert_close(s["rays_d"].cpu(), so["rays_d"], rtol=1e-6, what="rays_d")
    assert_close(s["z_vals"].cpu(), so["z_vals"], rtol=1e-6, what="z_vals")
    assert_close(s["pts"].cpu(), so["pts"], rtol=1e-6, what="pts")
    # class-balanced part: every class of each frame is drawn (select_by_class, utils/common.py:313-328)
    n1, n2 = prep["n1"], prep["n2"]
    lab = s["gt_label"].cpu()
    assert lab.numel() == 4 * (n1 + n2)
    for f in range(4):
        by_class = lab[f * npf + n1:(f + 1) * npf]
        present = torch.unique(frames["gt_label"][f]).long()
        assert set(torch.unique(by_class).tolist()) == set(present.tolist())


@pytest.mark.parametrize("nn,nl,layout", [(32, 1, "reference_tiled"), (64, 2, "reference_tiled"), (32, 1, "per_ray")])
def test_mapper_renderer_loss_and_gradients(nn, nl, layout):
    cfg, bound, cam, frames, dec, mapper = _setup(nn, nl, layout=layout)
    mapper.is_BA = True
    _, quad_list, T_list = mapper.set_optimizer(frames)
    prep = mapper.prepare_frames(frames)
    torch.manual_seed(5)
    pix = mapper.draw_pixels(prep)
    jit = mapper.draw_jitter()
    g = torch.Generator().manual_seed(6)
    u_off, u_jit = torch.rand(3, generator=g), torch.rand((1, 1, 1, 3), generator=g)
    samples = mapper.get_target_samples(frames, quad_list, T_list, prep=prep, pix_idx=pix, jitter=jit)
    N, S = samples["z_vals"].shape
    feats = torch.rand(N, S, 32, generator=g)
    samples["features"] = feats.to(DEV)
    loss, terms = mapper.iteration_loss(samples, lambda_lt=10.0, smooth=True, u_offset=u_off, u_jitter=u_jit, strict=True)
    loss.backward()

    om = oracle_from_product(cfg, bound, dec, mapper)
    qo = [q.detach().cpu().clone().requires_grad_(q.requires_grad) for q in quad_list]
    To = [t.detach().cpu().clone().requires_grad_(t.requires_grad) for t in T_list]
    npf = pix.numel() // 4
    so = _oracle_samples(frames, qo, To, cam, bound, pix.cpu(), (jit[0].cpu(), jit[1].cpu()), npf, 32, 15)
    so["features"] = feats
    lc = sr.LossCfg(smooth_pts=cfg["training"]["smooth_pts"])
    lo, to, outs = sr.mapping_loss(om, so, lc, u_off, u_jit, label_layout=layout)
    lo.backward()

    pc, pd, pv, pl, fine, coarse = mapper.renderer(samples)
    assert_close(pc.cpu(), outs["rgb"], what="pred_color")
    assert_close(pd.cpu(), outs["depth"], what="pred_depth")
    assert_close(pv.cpu(), outs["var"], what="pred_depth_var")
    assert_close(pl.cpu(), outs["logits"], what="pred_logits")
    assert_close(fine.cpu(), outs["fine"], what="fine_latents")
    assert_close(coarse.cpu(), outs["coarse"], what="coarse_latents")
    for k_p, k_o in (("p_loss", "p"), ("d_loss", "d"), ("l_loss", "l"), ("lt_loss", "lt"), ("fs_loss", "fs"),
                     ("opacity_loss", "op"), ("smooth_loss", "sm")):
        a, b = float(terms[k_p]), float(to[k_o])
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-6), f"{k_p}: {a} vs {b}"
    assert abs(float(loss) - float(lo)) <= 1e-4 * abs(float(lo))

    assert_close(dec.pe_fn.grid_fn.params.grad.cpu().reshape(-1, 2), om.table.grad, what="d table", groups=table_level_groups(om.meta))
    used = lambda n_in, n_out: nn * n_in + (nl - 1) * nn * nn + n_out * nn
    assert_close(dec.coarse_fn.decoder.params.grad.cpu()[:used(80, 33)], om.coarse.grad[:used(80, 33)], what="d coarse")
    assert_close(dec.out_fn.color_decoder.params.grad.cpu()[:used(112, 3)], om.color.grad[:used(112, 3)], what="d color")
    assert_close(dec.out_fn.logit_decoder.params.grad.cpu()[:used(112, 8)], om.logit.grad[:used(112, 8)], what="d logit")
    pool_grad = mapper.fine_decoders.pool.grad.cpu()
    for c, slot in mapper.fine_decoders.slot.items():
        go = om.fine[c].grad
        if go is None:
            assert torch.count_nonzero(pool_grad[slot]) == 0
        else:
            assert_close(pool_grad[slot][:used(80, 33)], go[:used(80, 33)], what=f"d fine[{c}]")
    for f in range(1, 4):                                 # frame 0 is fixed (slams/mapping.py:457)
        assert_pose_grad_close(quad_list[f], quad_list[f].grad, qo[f].grad, T_list[f].grad, To[f].grad, what=f"frame {f}")
    assert quad_list[0].grad is None


def test_fine_fn_unknown_class_raises():
    cfg, bound, cam, frames, dec, mapper = _setup()
    pe = torch.rand(256, 48, device=DEV)
    grid = torch.rand(256, 32, device=DEV)
    classes = torch.full((256,), 31, device=DEV, dtype=torch.int64)      # no decoder for class 31
    with pytest.raises(ValueError):
        mapper.fine_fn(pe, classes=classes, features=grid)


def test_tracker_renderer_losses_and_pose_gradient():
    from dns_slam_amd.tracking import Tracker
    cfg, bound, cam, frames, dec, mapper = _setup()
    cfg["tracking"]["n_pixels"] = 200
    tracker = Tracker(cfg, dec, bound, cam, device=DEV)
    tracker.border = 5
    cur = {"gt_color": frames["gt_color"][1], "gt_depth": frames["gt_depth"][1], "gt_label": frames["gt_label"][1]}
    _, quad, T = tracker.set_optimizer(frames["est_c2w"][1])
    cur["est_quad"], cur["est_T"] = quad, T
    torch.manual_seed(8)
    pix, jit = tracker.draw_pixels(), tracker.draw_jitter()
    s = tracker.get_target_samples(cur, pix_idx=pix, jitter=jit)
    g = torch.Generator().manual_seed(9)
    feats = torch.rand(200, 47, 32, generator=g)
    s["features"] = feats.to(DEV)
    pc, pd, pv, pl = tracker.renderer(s)
    loss = 5.0 * tracker.compute_photometric_loss(s["gt_color"], pc, s["mask"]) \
        + 5.0 * tracker.compute_depth_loss(s["gt_depth"], pd, pv, s["mask"]) \
        + 0.1 * tracker.compute_label_loss(s["gt_label"], pl, s["mask"])
    loss.backward()

    om = oracle_from_product(cfg, bound, dec)
    qo, To = quad.detach().cpu().clone().requires_grad_(True), T.detach().cpu().clone().requires_grad_(True)
    img5 = torch.cat((cur["gt_color"], cur["gt_depth"][..., None], cur["gt_label"][..., None]), -1)
    b = tracker.border
    so = sr.frame_samples(img5, qo, To, (cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]), bound, pix.cpu(),
                          jit[0].cpu(), jit[1].cpu(), 32, 15, window=(b, cam["H"] - b, b, cam["W"] - b))
    so["features"] = feats
    mask = (so["gt_depth"] > 0.01) & so["inside"]
    assert torch.equal(mask, s["mask"].cpu())
    lo, _, outs = sr.tracking_loss(om, so, mask)
    lo.backward()
    assert_close(pc.cpu(), outs["rgb"], what="color")
    assert_close(pd.cpu(), outs["depth"], what="depth")
    assert_close(pv.cpu(), outs["var"], what="var")
    assert_close(pl.cpu(), outs["logits"], what="logits")
    assert abs(float(loss) - float(lo)) <= 1e-4 * abs(float(lo))
    assert_pose_grad_close(quad, quad.grad, qo.grad, T.grad, To.grad, what="tracker")


def test_mapping_iterations_reduce_loss():
    """Integration: a few optimise iterations on the analytic room lower the loss (reference has no such test)."""
    cfg, bound, cam, frames, dec, mapper = _setup(n_pixels=800)
    from dns_slam_amd.decoder import Decoder
    dec2 = Decoder(cfg["model"], bound, n_class=8).to(DEV)
    from dns_slam_amd.mapping import Mapper
    m2 = Mapper(cfg, dec2, bound, cam, device=DEV)
    torch.manual_seed(0)
    m2.set_decoder(frames)
    opt, ql, Tl = m2.set_optimizer(frames)
    for gidx in range(3):
        opt.param_groups[gidx]["lr"] = 0.005 if gidx == 0 else 0.0
    prep = m2.prepare_frames(frames)
    losses = []
    for it in range(40):
        opt.zero_grad()
        s = m2.get_target_samples(frames, ql, Tl, prep=prep)
        loss, _ = m2.iteration_loss(s, smooth=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert sum(losses[-5:]) / 5 < 0.6 * (sum(losses[:5]) / 5), losses


def test_decoder_module_contract():
    """.state_dict() keys, deepcopy, pickle, share_memory-free CUDA sharing surface."""
    import copy
    import pickle
    cfg, bound, cam, frames, dec, mapper = _setup()
    keys = set(dec.state_dict().keys())
    for k in ("pe_fn.grid_fn.params", "coarse_fn.decoder.params", "out_fn.color_decoder.params",
              "out_fn.logit_decoder.params", "merge.decoder.params"):
        assert k in keys
    d2 = copy.deepcopy(dec)
    d3 = pickle.loads(pickle.dumps(dec))
    x = torch.rand(300, 3, device=DEV)
    for d in (d2, d3):
        pe, gr = d.pe_fn(x)
        pe0, gr0 = dec.pe_fn(x)
        assert torch.equal(pe, pe0) and torch.equal(gr, gr0)
        assert torch.equal(d.coarse_fn(pe, features=gr), dec.coarse_fn(pe0, features=gr0))
    assert dec.pe_dim == 48 and dec.grid_dim == 32
    col, log = dec.out_fn(pe0, torch.rand(300, 64, device=DEV))
    assert col.shape == (300, 3) and log.shape == (300, 8) and float(col.min()) >= 0 and float(col.max()) <= 1


def _loss_and_grads(mapper, dec, samples, u_off, u_jit):
    for p in list(dec.parameters()) + [mapper.fine_decoders.pool]:
        p.grad = None
    loss, terms = mapper.iteration_loss(samples, lambda_lt=10.0, smooth=True, u_offset=u_off, u_jitter=u_jit)
    loss.backward()
    grads = [dec.pe_fn.grid_fn.params.grad.clone(), dec.coarse_fn.decoder.params.grad.clone(),
             dec.out_fn.logit_decoder.params.grad.clone(), mapper.fine_decoders.pool.grad.clone()]
    return float(loss), {k: float(v) for k, v in terms.items()}, grads


def test_fused_losses_match_torch_losses():
    """csrc/losses.hip vs the per-term torch formulas (compute_*_loss + get_opacity_loss), values and gradients."""
    cfg, bound, cam, frames, dec, mapper = _setup()
    mapper.is_BA = False
    _, ql, Tl = mapper.set_optimizer(frames)
    torch.manual_seed(21)
    s = mapper.get_target_samples(frames, ql, Tl)
    g = torch.Generator().manual_seed(22)
    u_off, u_jit = torch.rand(3, generator=g), torch.rand((1, 1, 1, 3), generator=g)
    mapper.fused_losses = True
    lf, tf, gf = _loss_and_grads(mapper, dec, s, u_off, u_jit)
    mapper.fused_losses = False
    lu, tu, gu = _loss_and_grads(mapper, dec, s, u_off, u_jit)
    assert abs(lf - lu) <= 1e-5 * abs(lu)
    for k in tu:
        assert abs(tf[k] - tu[k]) <= 1e-5 * max(abs(tu[k]), 1e-7), k
    for a, b in zip(gf, gu):
        assert_close(a.cpu(), b.cpu(), rtol=1e-5, what="fused vs torch loss gradient")


@pytest.mark.parametrize("nn,nl", [(32, 1), (64, 2)])
def test_fused_render_nets_match_per_network_path(nn, nl):
    """ops.render_nets (two-segment colour / logit input, in-place input-gradient sums) vs the module-by-module path with
    torch.cat and autograd's own gradient accumulation: same loss, same gradients (incl. poses through the samples)."""
    cfg, bound, cam, frames, dec, mapper = _setup(n_neurons=nn, n_hidden_layers=nl)
    mapper.is_BA = True
    _, ql, Tl = mapper.set_optimizer(frames)
    torch.manual_seed(31)
    prep = mapper.prepare_frames(frames)
    pix, jit = mapper.draw_pixels(prep), mapper.draw_jitter()
    g = torch.Generator().manual_seed(32)
    u_off, u_jit = torch.rand(3, generator=g), torch.rand((1, 1, 1, 3), generator=g)
    res = []
    for fused in (True, False):
        mapper.fused_nets = fused
        for q in ql + Tl:
            q.grad = None
        s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=pix, jitter=jit)
        lo, terms, grads = _loss_and_grads(mapper, dec, s, u_off, u_jit)
        grads += [dec.out_fn.color_decoder.params.grad.clone()] + [q.grad.clone() for q in ql[1:] + Tl[1:]]
        res.append((lo, terms, grads))
    (lf, tf, gf), (lu, tu, gu) = res
    assert abs(lf - lu) <= 1e-5 * abs(lu)
    for k in tu:
        assert abs(tf[k] - tu[k]) <= 1e-5 * max(abs(tu[k]), 1e-7), k
    for a, b in zip(gf, gu):
        assert_close(a.cpu(), b.cpu(), rtol=2e-5, what="fused vs per-network gradient")


def test_smoothness_branch_on_a_second_stream_matches_single_stream():
    """Mapper.overlap_smooth: the lattice branch runs (forward and backward) on a side stream; same loss and gradients."""
    cfg, bound, cam, frames, dec, mapper = _setup()
    mapper.is_BA = False
    _, ql, Tl = mapper.set_optimizer(frames)
    torch.manual_seed(41)
    s = mapper.get_target_samples(frames, ql, Tl)
    g = torch.Generator().manual_seed(42)
    u_off, u_jit = torch.rand(3, generator=g), torch.rand((1, 1, 1, 3), generator=g)
    res = []
    for ov in (True, False, True):
        mapper.overlap_smooth = ov
        res.append(_loss_and_grads(mapper, dec, s, u_off, u_jit))
        torch.cuda.synchronize()
    for lo, terms, grads in (res[0], res[2]):
        assert abs(lo - res[1][0]) <= 1e-6 * abs(res[1][0])
        for a, b in zip(grads, res[1][2]):
            assert_close(a.cpu(), b.cpu(), rtol=1e-5, what="two-stream vs one-stream gradient")


def test_mapper_iteration_in_fp16_mlp_mode_tracks_fp32():
    """cfg['model']['mlp']['dtype'] = 'fp16' (BASELINE configs[4]): same scene, parameters and rays as the fp32 path; the
    loss terms agree to 2e-2 and training still reduces the loss (looser, stated tolerance)."""
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.mapping import Mapper
    out = {}
    for dtype in ("fp32", "fp16"):
        cam = synthetic.camera(H=60, W=80, fx=60.0, fy=60.0)
        bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=0)
        cfg = synthetic.default_cfg(n_pixels=360, n_samples_ray=32, n_surface_ray=15, n_frames=4, hash_size=14, voxel_size=0.08,
                                    n_neurons=64, n_hidden_layers=2, smooth_pts=12, mlp_dtype=dtype)
        dec = Decoder(cfg["model"], bound, n_class=8).to(DEV)
        mapper = Mapper(cfg, dec, bound, cam, device=DEV)
        mapper.set_decoder(frames)
        randomise_(dec, 11, scale=1.0)
        with torch.no_grad():
            dec.pe_fn.grid_fn.params.mul_(2000.0)
        randomise_([mapper.fine_decoders.pool], 12)
        assert dec.coarse_fn.decoder.fp16 == (dtype == "fp16") and mapper.fine_decoders.fp16 == (dtype == "fp16")
        mapper.is_BA = False
        opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
        for grp, lr in zip(opt.param_groups, (0.01, 0.0, 0.0)):
            grp["lr"] = lr
        torch.manual_seed(51)
        prep = mapper.prepare_frames(frames)
        pix, jit = mapper.draw_pixels(prep), mapper.draw_jitter()
        g = torch.Generator().manual_seed(52)
        u_off, u_jit = torch.rand(3, generator=g), torch.rand((1, 1, 1, 3), generator=g)
        hist = []
        for it in range(12):
            opt.zero_grad(set_to_none=True)
            s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=pix, jitter=jit)
            loss, terms = mapper.iteration_loss(s, smooth=True, u_offset=u_off, u_jitter=u_jit)
            loss.backward()
            opt.step()
            hist.append((float(loss), {k: float(v) for k, v in terms.items()}))
        out[dtype] = hist
    l32, t32 = out["fp32"][0]
    l16, t16 = out["fp16"][0]
    assert abs(l16 - l32) <= 2e-2 * abs(l32)
    for k in t32:
        assert abs(t16[k] - t32[k]) <= 2e-2 * max(abs(t32[k]), 1e-3), k
    assert out["fp16"][-1][0] < 0.9 * out["fp16"][0][0], "fp16 mode does not train"
    assert abs(out["fp16"][-1][0] - out["fp32"][-1][0]) <= 0.1 * abs(out["fp32"][-1][0])


def test_sync_free_smoothness_path_equals_reference_formulation():
    """static_shapes smoothness (fp64 affine lattice map, coarse network evaluated for its occupancy row only) vs the
    reference-ordered formulation (slams/mapping.py:133-157) on the same random offsets: value and gradients."""
    cfg, bound, cam, frames, dec, mapper = _setup(n_neurons=64, n_hidden_layers=2)
    ps = [dec.pe_fn.grid_fn.params, dec.coarse_fn.decoder.params]
    out = []
    for static in (True, False):
        for p in ps:
            p.grad = None
        mapper.static_shapes = static
        torch.cuda.manual_seed(77)
        if static:
            val = mapper.smoothness(sample_points=cfg["training"]["smooth_pts"])
        else:
            r = torch.rand(6, device=DEV)                      # what the sync-free path draws: [offset | jitter]
            val = mapper.smoothness(sample_points=cfg["training"]["smooth_pts"], u_offset=r[:3], u_jitter=r[3:].reshape(1, 1, 1, 3))
        val.backward()
        out.append((float(val), [p.grad.clone() for p in ps]))
    mapper.static_shapes = False
    assert abs(out[0][0] - out[1][0]) <= 1e-5 * abs(out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert_close(a.cpu(), b.cpu(), rtol=2e-5, what="sync-free vs reference-ordered smoothness gradient")


def test_static_shapes_masking_equals_ray_dropping():
    """Rays whose depth leaves the box: dropped by the reference (slams/mapping.py:576-586, host sync) vs kept with
    valid=0 in the sync-free path -- same loss and gradients (per_ray label layout)."""
    cfg, bound, cam, frames, dec, mapper = _setup(layout="per_ray")
    tight = bound.clone()
    tight[0, 1] -= 2.2                                   # pull one wall inside the room: some depths now exceed the box
    mapper.bound = tight
    mapper.bound_dev = tight.to(DEV)
    mapper.is_BA = False
    _, ql, Tl = mapper.set_optimizer(frames)
    prep = mapper.prepare_frames(frames)
    torch.manual_seed(31)
    pix, jit = mapper.draw_pixels(prep), mapper.draw_jitter()
    g = torch.Generator().manual_seed(32)
    u_off, u_jit = torch.rand(3, generator=g), torch.rand((1, 1, 1, 3), generator=g)
    mapper.static_shapes = False
    s_dyn = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=pix, jitter=jit)
    mapper.static_shapes = True
    s_sta = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=pix, jitter=jit)
    n_drop = int((s_sta["valid"] == 0).sum())
    assert 0 < n_drop < s_sta["valid"].numel() and s_dyn["z_vals"].shape[0] == s_sta["valid"].numel() - n_drop
    ld, td, gd = _loss_and_grads(mapper, dec, s_dyn, u_off, u_jit)
    ls, ts, gs = _loss_and_grads(mapper, dec, s_sta, u_off, u_jit)
    assert abs(ld - ls) <= 1e-5 * abs(ld), (ld, ls)
    for a, b in zip(gs, gd):
        assert_close(a.cpu(), b.cpu(), rtol=1e-4, what="masked vs dropped gradient")


def test_tracker_fused_losses_match_torch_losses():
    from dns_slam_amd import ops
    from dns_slam_amd.tracking import Tracker
    cfg, bound, cam, frames, dec, mapper = _setup()
    tracker = Tracker(cfg, dec, bound, cam, device=DEV)
    g = torch.Generator().manual_seed(41)
    N, C = 300, 8
    mk = lambda *sh: torch.rand(*sh, generator=g).to(DEV)
    pc, pd, pv, pl = mk(N, 3).requires_grad_(True), (mk(N) * 3).requires_grad_(True), (mk(N) + 0.01).requires_grad_(True), \
        (mk(N, C) * 4).requires_grad_(True)
    gc, gd_ = mk(N, 3), mk(N) * 3
    gl = torch.randint(0, C, (N,), generator=g).to(DEV)
    mask = (torch.rand(N, generator=g) < 0.8).to(DEV)
    loss_f, terms = ops.tracking_losses(pc, pd, pv, pl, gc, gd_, gl, mask, (5.0, 5.0, 0.1))
    gf = torch.autograd.grad(loss_f, (pc, pd, pv, pl))
    loss_t = 5.0 * tracker.compute_photometric_loss(gc, pc, mask) + 5.0 * tracker.compute_depth_loss(gd_, pd, pv, mask) \
        + 0.1 * tracker.compute_label_loss(gl, pl, mask)
    gt = torch.autograd.grad(loss_t, (pc, pd, pv, pl))
    assert abs(float(loss_f) - float(loss_t)) <= 1e-5 * abs(float(loss_t))
    for a, b in zip(gf, gt):
        assert_close(a.cpu(), b.cpu(), rtol=1e-5, what="tracker fused loss gradient")


def test_graph_capture_of_full_iteration():
    """The sync-free iteration (static_shapes) replays from a hipGraph and keeps training."""
    cfg, bound, cam, frames, dec, mapper = _setup(n_pixels=400)
    mapper.static_shapes = True
    mapper.is_BA = True
    opt, ql, Tl = mapper.set_optimizer(frames, capturable=True)
    for grp, lr in zip(opt.param_groups, (0.005, 0.0005, 0.0005)):
        grp["lr"] = torch.tensor(lr, device=DEV)
    prep = mapper.prepare_frames(frames)
    losses = []

    def step():
        opt.zero_grad(set_to_none=True)
        s = mapper.get_target_samples(frames, ql, Tl, prep=prep)
        loss, _ = mapper.iteration_loss(s, smooth=True)
        loss.backward()
        opt.step()
        return loss

    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = step()
    for _ in range(30):
        gr.replay()
        losses.append(float(out))
    assert all(l == l for l in losses)                      # finite
    assert len(set(losses)) > 20                            # fresh rays every replay (graph-safe RNG), not a frozen batch
    assert sum(losses[-5:]) < sum(losses[:5])


def test_render_frame_matches_oracle_chunks():
    """Full-image render (frame_vis, slams/mapping.py:638-724): whole-image sampling + chunked renderer vs the oracle."""
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.mapping import Mapper
    cam = synthetic.camera(H=24, W=32, fx=24.0, fy=24.0)
    bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=3)
    cfg = synthetic.default_cfg(n_pixels=200, hash_size=14, voxel_size=0.08, smooth_pts=10)
    dec = Decoder(cfg["model"], bound, n_class=8).to(DEV)
    mapper = Mapper(cfg, dec, bound, cam, device=DEV)
    mapper.set_decoder(frames)
    randomise_(dec, 5)
    with torch.no_grad():
        dec.pe_fn.grid_fn.params.mul_(2000.0)
    randomise_([mapper.fine_decoders.pool], 6)
    torch.manual_seed(1)
    jit = mapper.draw_jitter(1)                       # one frame: [1, n_surface]
    c2w = frames["est_c2w"][2]
    col, dep, lab = mapper.render_frame(frames["gt_color"][2], frames["gt_depth"][2], frames["gt_label"][2], c2w,
                                        n_pts_batch=200, jitter=jit)
    om = oracle_from_product(cfg, bound, dec, mapper)
    from dns_slam_amd.common import get_quad_from_c2w
    img5 = torch.cat((frames["gt_color"][2], frames["gt_depth"][2][..., None], frames["gt_label"][2][..., None]), -1)
    so = sr.frame_samples(img5, get_quad_from_c2w(c2w), c2w[:3, 3].clone(), (24, 32, 24.0, 24.0, cam["cx"], cam["cy"]), bound,
                          torch.arange(24 * 32), jit[0][0].cpu(), jit[1][0].cpu(), 32, 15)
    cols, deps, labs = [], [], []
    for st in range(0, 24 * 32, 200):
        chunk = {k: so[k][st:st + 200] for k in ("pts", "z_vals", "gt_label", "features")}
        rgb, depth, _, logits, _, _ = sr.mapper_renderer(om, chunk)
        cols.append(rgb.detach()), deps.append(depth.detach()), labs.append(torch.argmax(logits, -1))
    assert_close(col.cpu().reshape(-1, 3), torch.cat(cols), what="render_frame colour")
    assert_close(dep.cpu().reshape(-1), torch.cat(deps), what="render_frame depth")
    agree = (lab.cpu().reshape(-1) == torch.cat(labs)).float().mean()
    assert float(agree) > 0.995          # argmax may flip where two logits tie to 1e-4


def test_eval_points_matches_oracle():
    """Meshing / evaluation query (slams/meshing.py:461-503): colour + occupancy of arbitrary world points, -100 outside
    the open bound, fine-decoder routing by label with the > 1 point rule, argmax labels; chunked evaluation."""
    cfg, bound, cam, frames, dec, mapper = _setup()
    om = oracle_from_product(cfg, bound, dec, mapper)
    g = torch.Generator().manual_seed(8)
    P = 3000
    ext = (bound[:, 1] - bound[:, 0]).float()
    pts = (torch.rand(P, 3, generator=g) * 1.2 - 0.1) * ext + bound[:, 0].float()        # ~40 % outside
    pix = torch.rand(P, cfg["model"]["hidden_dim"], generator=g) * 2 - 1
    known = sorted(mapper.fine_decoders.keys())
    lab = torch.tensor(known)[torch.randint(0, len(known) - 1, (P,), generator=g)]
    lab[17] = known[-1]                                   # a class with exactly one point -> zeros from the fine stage
    for stage in ("coarse", "fine"):
        vp, lp = mapper.eval_points(pts, pixel_pts=pix, gt_label_pts=lab, stage=stage, n_pts_batch=1024)
        vo, lo = sr.eval_points(om, pts, pix, lab, stage)
        assert_close(vp.cpu(), vo.detach(), what=f"eval_points values ({stage})")
        inside = vo[:, 3] != -100
        assert 0.3 < float(inside.float().mean()) < 0.9
        if stage == "fine":
            agree = (lp.cpu() == lo).float().mean()
            assert float(agree) > 0.995                   # argmax may flip where two logits tie to 1e-4
            assert bool((lp.cpu()[~inside] == -1).all())
            assert float(vp[17, 3]) in (0.0, -100.0)      # single-point class: occupancy logit 0 (or outside)
        else:
            assert lp is None
        # without a 2-D code (the marching-cubes driver's call): the colour / logit networks run on their live inputs only
        vz, lz = mapper.eval_points(pts, pixel_pts=None, gt_label_pts=lab, stage=stage, n_pts_batch=1024)
        voz, loz = sr.eval_points(om, pts, torch.zeros_like(pix), lab, stage)
        assert_close(vz.cpu(), voz.detach(), what=f"eval_points values without a code ({stage})")
        if stage == "fine":
            assert float((lz.cpu() == loz).float().mean()) > 0.995
    lab[5] = 999
    with pytest.raises(ValueError):
        mapper.eval_points(pts, pixel_pts=pix, gt_label_pts=lab, stage="fine")


def test_tracker_track_frame_eager_and_graphed_reduce_pose_error():
    """Per-frame tracking loop (slams/tracking.py:313-340): from a perturbed pose, both the eager loop and the
    hipGraph-replayed loop lower the loss and return the best-loss camera."""
    from dns_slam_amd.tracking import Tracker
    from dns_slam_amd.common import get_camera_from_tensor
    cfg, bound, cam, frames, dec, mapper = _setup(n_pixels=800)
    # a few mapping iterations so that the scene carries signal
    opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
    for gi, lr in enumerate((0.005, 0.0, 0.0)):
        opt.param_groups[gi]["lr"] = lr
    prep = mapper.prepare_frames(frames)
    torch.manual_seed(0)
    for _ in range(60):
        opt.zero_grad()
        s = mapper.get_target_samples(frames, ql, Tl, prep=prep)
        loss, _ = mapper.iteration_loss(s, smooth=False)
        loss.backward()
        opt.step()
    cfg["tracking"]["n_pixels"] = 300
    tracker = Tracker(cfg, dec, bound, cam, device=DEV)
    tracker.border = 5
    cur = {"gt_color": frames["gt_color"][2], "gt_depth": frames["gt_depth"][2], "gt_label": frames["gt_label"][2]}
    c2w = frames["est_c2w"][2].clone()
    c2w[:3, 3] += torch.tensor([0.03, -0.02, 0.02])
    for graph in (False, True):
        torch.manual_seed(1)
        cam7, best = tracker.track_frame(cur, c2w, n_iters=40, fused=True, graph=graph)
        assert cam7.shape == (7,) and bool(torch.isfinite(cam7).all()) and float(best) == float(best)
        err0 = float((c2w[:3, 3] - frames["est_c2w"][2][:3, 3]).norm())
        err1 = float((cam7[4:].cpu() - frames["est_c2w"][2][:3, 3]).norm())
        assert err1 < err0 * 1.5                 # does not diverge (40 tiny-lr steps move the pose by <= 0.04)


def test_tracker_graph_replay_equals_eager_loop():
    """Tracker.track_frame(graph=True) -- one captured iteration replayed n_iters times -- against the eager loop with the
    same device-side draws (static_shapes) from the same seed: the same keep-best loss (1e-5) and camera (2e-5: float atomics in the
    pose-gradient reduction are the only run-to-run difference, amplified over 25 Adam steps; Philox offsets advance per
    replay exactly as the eager generator does)."""
    from dns_slam_amd.tracking import Tracker
    cfg, bound, cam, frames, dec, mapper = _setup(64, 2, n_pixels=400)
    cfg["tracking"]["n_pixels"] = 256
    cur = {"gt_color": frames["gt_color"][2], "gt_depth": frames["gt_depth"][2], "gt_label": frames["gt_label"][2]}
    c2w = frames["est_c2w"][2].clone()
    c2w[:3, 3] += torch.tensor([0.02, -0.01, 0.015], dtype=c2w.dtype)
    out = {}
    for graph in (False, True):
        tracker = Tracker(cfg, dec, bound, cam, device=DEV)
        tracker.border = 5
        tracker.static_shapes = True
        torch.manual_seed(3)
        cam7, best = tracker.track_frame(cur, c2w, n_iters=25, fused=True, graph=graph)
        out[graph] = (cam7.detach().cpu().clone(), float(best))
    assert abs(out[True][1] - out[False][1]) <= 1e-5 * abs(out[False][1]), (out[True][1], out[False][1])
    assert float((out[True][0] - out[False][0]).abs().max()) <= 2e-5, (out[True][0], out[False][0])


def test_graph_replay_survives_host_synchronisation():
    """Regression for the round-2 finding: memset nodes of a captured hipGraph stop clearing their
    destination once a host synchronisation has followed a replay (ROCm 7.0.51831), so the library fills by kernel.  A whole
    Mapper iteration with FIXED draws and no optimiser step, captured and replayed as 3 x replay + read, then 4 x (3 x
    replay, torch.cuda.synchronize(), read): every read must give the same loss (float atomics: 1e-6)."""
    cfg, bound, cam, frames, dec, mapper = _setup(64, 2, n_pixels=400)
    mapper.static_shapes = True
    mapper.is_BA = True
    opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
    prep = mapper.prepare_frames(frames)
    torch.manual_seed(5)
    fixed = (mapper.draw_pixels(prep), mapper.draw_jitter(), torch.rand(3, device=DEV), torch.rand((1, 1, 1, 3), device=DEV))
    params = [p for g in opt.param_groups for p in g["params"]]

    def fn():
        for p in params:
            p.grad = None
        s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=fixed[0], jitter=fixed[1])
        loss, _ = mapper.iteration_loss(s, smooth=True, u_offset=fixed[2], u_jitter=fixed[3])
        loss.backward()
        return loss + 0.0 * sum(p.grad.abs().sum() for p in params if p.grad is not None)

    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        out = fn()
    vals = []
    for _ in range(3):
        gr.replay()
    vals.append(float(out.detach()))
    for _ in range(4):
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        vals.append(float(out.detach()))
    assert all(v == v for v in vals) and max(vals) - min(vals) <= 1e-6 * abs(vals[0]), vals


def test_optimize_driver_and_decoder_init():
    """Mapper.optimize (slams/mapping.py:839-949) end to end incl. the decoder warm-up (:764-836) for classes that appear
    after frame 50: the warm-up only draws rays of the new classes, the driver returns a valid pose and writes the
    refined keyframe poses back."""
    cfg, bound, cam, frames, dec, mapper = _setup(n_pixels=400)
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.mapping import Mapper
    dec = Decoder(cfg["model"], bound, n_class=8).to(DEV)
    m = Mapper(cfg, dec, bound, cam, device=DEV)
    fr = dict(frames)
    fr["est_c2w"] = frames["est_c2w"].clone()
    fr["label_dict"] = [0, 1, 2, 3]
    torch.manual_seed(0)
    c2w0, terms0 = m.optimize(6, 20, fr, smooth=True)                   # no warm-up before frame 50
    assert set(m.fine_decoders.keys()) == {0, 1, 2, 3}
    fr["label_dict"] = sorted(frames["label_dict"])                      # classes 4..7 appear
    pool_before = m.fine_decoders.pool.detach().clone()
    c2w1, terms1 = m.optimize(6, 60, fr, smooth=True)                   # > 50: decoder_init runs for the new classes
    assert set(m.fine_decoders.keys()) == set(frames["label_dict"])
    for c in (4, 5, 6, 7):
        s_ = m.fine_decoders.slot[c]
        assert float((m.fine_decoders.pool[s_] - pool_before[s_]).abs().max()) > 0       # warmed up / trained
    assert c2w1.shape == (4, 4) and bool(torch.isfinite(c2w1).all())
    R = c2w1[:3, :3]
    assert torch.allclose(R @ R.t(), torch.eye(3, device=DEV), atol=1e-4)
    assert all(float(v) == float(v) for v in terms1.values())
    assert not torch.equal(fr["est_c2w"][1], frames["est_c2w"][1])      # BA pose write-back (idx >= start_optimize_idx)
    assert torch.equal(fr["est_c2w"][0], frames["est_c2w"][0])          # the oldest frame stays fixed (:457)


def test_optimise_trajectory_matches_oracle_adam():
    """Five complete optimise iterations (sample -> render -> seven losses -> backward -> Adam on grid, every network, the
    per-class fine decoders and the poses of frames 1-3; slams/mapping.py:868-911) of the HIP path with its fused Adam
    against the oracle stepped by torch.optim.Adam on the SAME pixel / jitter / lattice draws: the loss of every
    iteration within 1e-4 relative -- iteration k sees the parameters k updates produced -- and the poses after the run.
    Parameters themselves are compared through the loss: an element whose gradient is rounding noise moves by +-lr under
    Adam's normalisation in either implementation."""
    cfg, bound, cam, frames, dec, mapper = _setup(64, 2)
    mapper.is_BA = True
    opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
    lr, cam_lr = 2e-3, 1e-3
    for grp, v in zip(opt.param_groups, (lr, cam_lr, cam_lr)):
        grp["lr"] = v
    prep = mapper.prepare_frames(frames)

    om = oracle_from_product(cfg, bound, dec, mapper)
    qo = [q.detach().cpu().clone().requires_grad_(q.requires_grad) for q in ql]
    To = [t.detach().cpu().clone().requires_grad_(t.requires_grad) for t in Tl]
    opt_o = torch.optim.Adam([{"params": [om.table, om.coarse, om.color, om.logit] + list(om.fine.values()), "lr": lr},
                              {"params": qo[1:], "lr": cam_lr}, {"params": To[1:], "lr": cam_lr}])
    lc = sr.LossCfg(smooth_pts=cfg["training"]["smooth_pts"])
    for it in range(5):
        torch.manual_seed(100 + it)
        pix, jit = mapper.draw_pixels(prep), mapper.draw_jitter()
        g = torch.Generator().manual_seed(200 + it)
        u_off, u_jit = torch.rand(3, generator=g), torch.rand((1, 1, 1, 3), generator=g)

        opt.zero_grad()
        s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=pix, jitter=jit)
        loss, _ = mapper.iteration_loss(s, lambda_lt=10.0, smooth=True, u_offset=u_off, u_jitter=u_jit, strict=True)
        loss.backward()
        opt.step()

        opt_o.zero_grad()
        npf = pix.numel() // 4
        so = _oracle_samples(frames, qo, To, cam, bound, pix.cpu(), (jit[0].cpu(), jit[1].cpu()), npf, 32, 15)
        so["features"] = torch.zeros(so["z_vals"].shape[0], so["z_vals"].shape[1], 32)
        lo, _, _ = sr.mapping_loss(om, so, lc, u_off, u_jit, label_layout="reference_tiled")
        lo.backward()
        opt_o.step()
        a, b = float(loss.detach()), float(lo.detach())
        assert abs(a - b) <= 1e-4 * abs(b), f"iteration {it}: {a} vs {b}"
    for f in range(1, 4):
        assert_close(ql[f].detach().cpu(), qo[f].detach(), rtol=1e-4, what=f"quat[{f}] after 5 steps")
        assert_close(Tl[f].detach().cpu(), To[f].detach(), rtol=1e-4, what=f"T[{f}] after 5 steps")



def test_optimize_reference_signature_with_keyframes_and_stem():
    """Mapper.optimize(n_iters, idx, color, depth, label, gt_c2w, cur_c2w) -> (c2w, loss dict): the reference's signature
    (slams/mapping.py:839).  Keyframes come from mapper.keyframe_dict, the frozen stem feeds the 2-D branch in every
    iteration, refined keyframe poses are written back (:914-926), and the same call through optimize_frames on the same
    bundle gives the same result (the adapter adds no arithmetic)."""
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.encoder import ResNet
    from dns_slam_amd.mapping import Mapper
    cfg, bound, cam, frames, _, _ = _setup(n_pixels=400)
    cfg["mapping"]["start_optimize_idx"] = 0
    run = {}
    for mode in ("reference", "frames"):
        torch.manual_seed(3)
        dec = Decoder(cfg["model"], bound, n_class=8).to(DEV)
        m = Mapper(cfg, dec, bound, cam, device=DEV)
        m.encoder = ResNet(seed=1).to(DEV)
        m.keyframe_selector = lambda color, depth, c2w, kfs, k: [1]     # host-side keyframe choice (out of scope), fixed here
        for k in range(3):                                   # three stored keyframes, the fourth frame is the current one
            m.keyframe_list.append(5 * k)
            m.keyframe_dict.append({"gt_color": frames["gt_color"][k], "gt_depth": frames["gt_depth"][k],
                                    "gt_label": frames["gt_label"][k], "gt_c2w": frames["gt_c2w"][k],
                                    "est_c2w": frames["est_c2w"][k].clone()})
        cur = (frames["gt_color"][3].to(DEV), frames["gt_depth"][3].to(DEV), frames["gt_label"][3].to(DEV),
               frames["gt_c2w"][3].to(DEV), frames["est_c2w"][3].to(DEV))
        if mode == "reference":
            c2w, out = m.optimize(4, 20, *cur)
            assert set(out) == {"loss_camera_tensor", "p_loss", "d_loss", "l_loss", "lt_loss", "smooth_loss"}
            assert m.n_target_frame == 3                       # keyframes 1, 2 (0 never joins, :361) + the current frame
            assert not torch.equal(m.keyframe_dict[2]["est_c2w"].cpu(), frames["est_c2w"][2])   # refined pose written back
            assert torch.equal(m.keyframe_dict[0]["est_c2w"].cpu(), frames["est_c2w"][0])
        else:
            idx, tf, rf = m.set_target_refer_frames(*cur)
            assert idx == [1, 2, -1] and rf["gt_color"].shape[:2] == (3, 3) and rf["kf_idx"][-1] == [1, 2, -1]
            feats = m.encoder(rf["gt_color"])
            c2w, out = m.optimize_frames(4, 20, tf, features=feats, refer_frames=rf)
        assert c2w.shape == (4, 4) and bool(torch.isfinite(c2w).all())
        run[mode] = (c2w.cpu(), float(out["p_loss"]), float(out["d_loss"]))
    assert_close(run["reference"][0], run["frames"][0], rtol=1e-5, what="optimize adapter pose")
    assert abs(run["reference"][1] - run["frames"][1]) <= 1e-4 * abs(run["frames"][1])


def test_keyframe_selection_overlap_matches_the_reference_procedure():
    """Mapper.keyframe_selection_overlap (slams/mapping.py:171-236): the per-keyframe overlap fractions against a numpy
    restatement of the reference's loop on the SAME sampled rays (device generator re-seeded), the threshold / ordering /
    ``np.random.permutation`` selection against the same steps on those fractions, and mapping_mode = 'overlap' routing in
    set_target_refer_frames (th = 0.05, :353-356)."""
    import numpy as np
    from dns_slam_amd import common
    cfg, bound, cam, frames, dec, mapper = _setup()
    H, W, fx, fy, cx, cy = cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    kfs = [{"gt_color": frames["gt_color"][k], "gt_depth": frames["gt_depth"][k], "gt_label": frames["gt_label"][k],
            "gt_c2w": frames["gt_c2w"][k], "est_c2w": frames["est_c2w"][k].clone()} for k in range(3)]
    color, depth, c2w = frames["gt_color"][3].to(DEV), frames["gt_depth"][3].to(DEV), frames["est_c2w"][3].to(DEV)
    torch.manual_seed(11)
    np.random.seed(5)
    picked = mapper.keyframe_selection_overlap(color, depth, c2w, kfs, 2, th=0.05)
    got = mapper.last_overlap_percent
    # the reference's loop, restated in numpy on the same rays
    torch.manual_seed(11)
    img = torch.cat((color, depth.unsqueeze(-1)), -1)
    ro, rd, smp = common.get_samples(0, H, 0, W, 100, H, W, fx, fy, cx, cy, c2w[:3, :3], c2w[:3, -1], img, DEV)
    d = smp[:, -1].reshape(-1, 1).repeat(1, 16)
    t = torch.linspace(0.0, 1.0, steps=16, device=DEV)
    z_vals = d * 0.8 * (1.0 - t) + (d + 0.5) * t
    vertices = (ro[..., None, :] + rd[..., None, :] * z_vals[..., :, None]).reshape(-1, 3).cpu().numpy()
    K = np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])
    want = []
    for kf in kfs:
        w2c = np.linalg.inv(kf["est_c2w"].cpu().numpy())
        homo = np.concatenate([vertices, np.ones_like(vertices[:, :1])], axis=1).reshape(-1, 4, 1)
        cc = (w2c @ homo)[:, :3]
        cc[:, 0] *= -1
        uv = K @ cc
        z = uv[:, -1:] + 1e-5
        uv = (uv[:, :2] / z).astype(np.float32)
        m = (uv[:, 0] < W - 10) * (uv[:, 0] > 10) * (uv[:, 1] < H - 10) * (uv[:, 1] > 10)
        m = (m & (z[:, :, 0] < 0)).reshape(-1)
        want.append(m.sum() / uv.shape[0])
    want = np.array(want)
    assert np.abs(got - want).max() <= 2.0 / vertices.shape[0], (got, want)      # a point within rounding of an edge may flip
    np.random.seed(5)
    order = sorted(range(3), key=lambda i: got[i], reverse=True)
    sel = [i for i in order if got[i] > 0.05]
    assert picked == [int(i) for i in np.random.permutation(np.array(sel))[:2]]
    # routing: mapping_mode 'overlap' takes this path in set_target_refer_frames
    for k in range(3):
        mapper.keyframe_list.append(5 * k)
        mapper.keyframe_dict.append(kfs[k])
    mapper.mapping_mode = "overlap"
    torch.manual_seed(11)
    np.random.seed(5)
    idx, tf, rf = mapper.set_target_refer_frames(color, depth, frames["gt_label"][3].to(DEV), frames["gt_c2w"][3].to(DEV), c2w)
    assert idx[-1] == -1 and 2 in idx and 0 not in idx and all(i in (1, 2, -1) for i in idx)
