"""``MapStep`` (dns_slam_amd/fused_step.py: the mapping iteration as a fixed launch sequence over preallocated buffers) against
the autograd-driven iteration of ``Mapper.optimize_frames`` (itself held to the oracle by test_gpu_slam.py / test_gpu_cfg1.py):
same seed -> same draws -> the same losses every iteration and the same parameters after several Adam steps."""
import copy

import pytest
import torch

from test_gpu_slam import _setup
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(fused, code, n_iters, layout, nn=32, nl=1):
    from dns_slam_amd.fused_step import MapStep
    from dns_slam_amd.optim import FusedAdam  # noqa: F401  (set_optimizer(fused=True))
    cfg, bound, cam, frames, dec, mapper = _setup(nn, nl, layout=layout)
    mapper.static_shapes, mapper.is_BA, mapper.overlap_smooth, mapper.prefetch_draws = True, True, True, True
    opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
    for grp, lr in zip(opt.param_groups, (mapper.lr, mapper.BA_cam_lr, mapper.BA_cam_lr)):
        grp["lr"] = lr
    prep = mapper.prepare_frames(frames)
    feats = None
    if code:
        g = torch.Generator().manual_seed(5)
        npf = prep["n1"] + prep["n2"]
        feats = (torch.rand(4 * npf, 32 + 15, 32, generator=g) * 2 - 1).to(DEV)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    hist, grads = [], None
    pool = mapper.fine_decoders.pool
    if fused:
        ms = MapStep(mapper, frames, ql, Tl, prep=prep, features=feats)
        for i in range(n_iters):
            ms.step()
            total, terms = ms.losses()
            hist.append((float(total), {k: float(v) for k, v in terms.items()}))
            if i == 0:
                grads = {"table": ms.g_table, "coarse": ms.g_coarse, "color": ms.g_color, "logit": ms.g_logit,
                         "pool": ms.g_pool.view_as(pool), "quat": ms.g_quat.view(4, 4)[1:], "trans": ms.g_trans.view(4, 3)[1:]}
                grads = {k: v.detach().cpu().clone() for k, v in grads.items()}
        ms.write_back()
    else:
        for i in range(n_iters):
            opt.zero_grad(set_to_none=True)
            s = mapper.get_target_samples(frames, ql, Tl, prep=prep, features=feats)
            loss, terms = mapper.iteration_loss(s, lambda_lt=10.0, smooth=True)
            loss.backward()
            if i == 0:
                grads = {"table": dec.pe_fn.grid_fn.params.grad, "coarse": dec.coarse_fn.decoder.params.grad,
                         "color": dec.out_fn.color_decoder.params.grad, "logit": dec.out_fn.logit_decoder.params.grad,
                         "pool": pool.grad, "quat": torch.stack([q.grad for q in ql[1:]]),
                         "trans": torch.stack([t.grad for t in Tl[1:]])}
                grads = {k: v.detach().cpu().clone() for k, v in grads.items()}
            opt.step()
            hist.append((float(loss.detach()), {k: float(v) for k, v in terms.items()}))
    torch.cuda.synchronize()
    params = {"table": dec.pe_fn.grid_fn.params, "coarse": dec.coarse_fn.decoder.params,
              "color": dec.out_fn.color_decoder.params, "logit": dec.out_fn.logit_decoder.params,
              "pool": mapper.fine_decoders.pool, "quat": torch.stack([q.detach() for q in ql]),
              "trans": torch.stack([t.detach() for t in Tl])}
    return hist, {k: v.detach().cpu().clone() for k, v in params.items()}, grads, (mapper.lr, mapper.BA_cam_lr)


@pytest.mark.parametrize("code,layout,net", [(False, "reference_tiled", (32, 1)), (True, "reference_tiled", (32, 1)),
                                             (True, "per_ray", (64, 2))])
def test_map_step_equals_the_autograd_iteration(code, layout, net):
    n = 6
    ha, pa, ga, lrs = _run(False, code, n, layout, *net)
    hf, pf, gf, _ = _run(True, code, n, layout, *net)
    for i, ((la, ta), (lf, tf)) in enumerate(zip(ha, hf)):
        assert abs(la - lf) <= 1e-4 * abs(la), (i, la, lf)
        for k in ta:
            assert abs(ta[k] - tf[k]) <= 1e-4 * max(abs(ta[k]), 1e-6), (i, k, ta[k], tf[k])
    for k in ga:                                                 # the first iteration's gradients, before Adam touches anything
        assert_close(gf[k], ga[k], rtol=1e-4, elementwise=False, what=f"MapStep vs autograd: d {k}, iteration 1")
    for k in pa:
        # Adam's first steps move every weight by ~lr whatever the size of its gradient: where the gradient is rounding noise
        # around zero (table rows no sample touched strongly) the two runs' steps differ by a fraction of lr.  So: every
        # parameter within 1e-4 of the tensor's scale + 5 % of the distance n Adam steps can move it
        lr = lrs[1] if k in ("quat", "trans") else lrs[0]
        diff = (pf[k] - pa[k]).abs().max().item()
        assert diff <= 1e-4 * pa[k].abs().max().item() + 0.05 * lr * n, (k, diff)
    assert ha[-1][0] < ha[0][0]                                  # and it trains


def test_map_step_frozen_poses_and_single_frame():
    """is_BA False: no pose gradient, no d(grid)/dx buffer; the poses stay bit-equal."""
    from dns_slam_amd.fused_step import MapStep
    cfg, bound, cam, frames, dec, mapper = _setup()
    mapper.static_shapes, mapper.is_BA = True, False
    _, ql, Tl = mapper.set_optimizer(frames, fused=True)
    q0 = torch.stack([q.detach().clone() for q in ql])
    torch.manual_seed(9)
    torch.cuda.manual_seed(9)
    ms = MapStep(mapper, frames, ql, Tl)
    l0 = None
    for i in range(8):
        ms.step()
        if i == 0:
            l0 = float(ms.losses()[0])
    ms.write_back()
    assert ms.dydx is None and torch.equal(torch.stack([q.detach() for q in ql]), q0)
    assert float(ms.losses()[0]) < l0


def test_map_step_without_smoothness_and_single_stream():
    """smooth=False (decoder_init-style iterations) and the one-stream mode (no side stream, no prefetch: what a hipGraph capture
    of the step uses) against the autograd iteration."""
    from dns_slam_amd.fused_step import MapStep
    out = []
    for fused in (False, True):
        cfg, bound, cam, frames, dec, mapper = _setup()
        mapper.static_shapes, mapper.is_BA, mapper.overlap_smooth, mapper.prefetch_draws = True, True, False, False
        opt, ql, Tl = mapper.set_optimizer(frames, fused=True)
        for grp, lr in zip(opt.param_groups, (mapper.lr, mapper.BA_cam_lr, mapper.BA_cam_lr)):
            grp["lr"] = lr
        prep = mapper.prepare_frames(frames)
        torch.manual_seed(77)
        torch.cuda.manual_seed(77)
        hist = []
        ms = MapStep(mapper, frames, ql, Tl, prep=prep, smooth=False) if fused else None
        for _ in range(4):
            if fused:
                ms.step()
                hist.append(float(ms.losses()[0]))
            else:
                opt.zero_grad(set_to_none=True)
                s = mapper.get_target_samples(frames, ql, Tl, prep=prep)
                loss, _ = mapper.iteration_loss(s, lambda_lt=10.0, smooth=False)
                loss.backward()
                opt.step()
                hist.append(float(loss.detach()))
        out.append(hist)
    for a, b in zip(*out):
        assert abs(a - b) <= 1e-4 * abs(a), out


def test_map_step_single_frame_optimises_its_pose():
    """One target frame (the first mapping call, slams/mapping.py:447-455: with n_target_frame == 1 the only frame's pose IS
    optimised): MapStep against the autograd iteration."""
    from dns_slam_amd.fused_step import MapStep
    out = []
    for fused in (False, True):
        cfg, bound, cam, frames, dec, mapper = _setup()
        one = {k: (v[:1] if (torch.is_tensor(v) or isinstance(v, list)) else v) for k, v in frames.items()}
        mapper.n_target_frame = 1
        mapper.static_shapes, mapper.is_BA, mapper.overlap_smooth, mapper.prefetch_draws = True, True, True, True
        opt, ql, Tl = mapper.set_optimizer(one, fused=True)
        assert ql[0].requires_grad
        for grp, lr in zip(opt.param_groups, (mapper.lr, mapper.BA_cam_lr, mapper.BA_cam_lr)):
            grp["lr"] = lr
        prep = mapper.prepare_frames(one)
        torch.manual_seed(4)
        torch.cuda.manual_seed(4)
        hist = []
        ms = MapStep(mapper, one, ql, Tl, prep=prep) if fused else None
        for _ in range(4):
            if fused:
                ms.step()
                hist.append(float(ms.losses()[0]))
            else:
                opt.zero_grad(set_to_none=True)
                s = mapper.get_target_samples(one, ql, Tl, prep=prep)
                loss, _ = mapper.iteration_loss(s, lambda_lt=10.0, smooth=True)
                loss.backward()
                opt.step()
                hist.append(float(loss.detach()))
        if fused:
            ms.write_back()
        torch.cuda.synchronize()
        out.append((hist, ql[0].detach().cpu().clone(), Tl[0].detach().cpu().clone()))
    for a, b in zip(out[0][0], out[1][0]):
        assert abs(a - b) <= 1e-4 * abs(a), (out[0][0], out[1][0])
    assert float((out[0][1] - out[1][1]).abs().max()) <= 2e-5 and float((out[0][2] - out[1][2]).abs().max()) <= 2e-5
    cfg, bound, cam, frames, dec, mapper = _setup()
    from dns_slam_amd.common import get_quad_from_c2w
    assert float((out[1][1] - get_quad_from_c2w(frames["est_c2w"][0])).abs().max()) > 0          # and the pose did move


def test_map_step_with_kept_hidden_activations_is_the_same_step():
    """MapStep(keep_hidden=True): every network's forward keeps its hidden activations and the backward reads them back instead of
    recomputing them -- the same losses (the kernels' results are bit-identical; float atomics give the usual last-bit noise)."""
    from dns_slam_amd.fused_step import MapStep
    out = []
    for keep in (False, True):
        cfg, bound, cam, frames, dec, mapper = _setup(64, 2)
        mapper.static_shapes, mapper.is_BA, mapper.overlap_smooth, mapper.prefetch_draws = True, True, True, True
        _, ql, Tl = mapper.set_optimizer(frames, fused=True)
        torch.manual_seed(12)
        torch.cuda.manual_seed(12)
        ms = MapStep(mapper, frames, ql, Tl, keep_hidden=keep)
        hist = []
        for _ in range(5):
            ms.step()
            hist.append(float(ms.losses()[0]))
        out.append(hist)
    for a, b in zip(*out):
        assert abs(a - b) <= 2e-5 * abs(a), out


def test_optimize_frames_through_map_step():
    """``Mapper.optimize_frames`` with ``use_map_step``: the reference's driver (set_decoder, the lambda_lt schedule of
    slams/mapping.py:893-896, pose write-back :914-926) around the fixed launch sequence -- same result as the autograd loop."""
    out = []
    for use in (False, True):
        cfg, bound, cam, frames, dec, mapper = _setup()
        mapper.fine_decoders.slot.pop(max(mapper.fine_decoders.slot))            # one class gets its decoder in this call:
        mapper.exist_decoders.pop(max(mapper.exist_decoders))                    # lambda_lt = 0 for the first half
        mapper.fine_decoders._lut = None
        mapper.static_shapes, mapper.overlap_smooth, mapper.prefetch_draws, mapper.use_map_step = True, True, True, use
        frames = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in frames.items()}
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        c2w, terms = mapper.optimize_frames(8, 20, frames)
        torch.cuda.synchronize()
        out.append((c2w.cpu(), {k: float(v) for k, v in terms.items()}, dec.coarse_fn.decoder.params.detach().cpu().clone()))
    (ca, ta, pa), (cb, tb, pb) = out
    assert_close(cb, ca, rtol=2e-4, what="optimize_frames pose: MapStep vs autograd")
    for k in ta:
        assert abs(ta[k] - tb[k]) <= 2e-4 * max(abs(ta[k]), 1e-6), (k, ta[k], tb[k])
    assert float((pa - pb).abs().max()) <= 1e-4 * float(pa.abs().max()) + 0.05 * 8 * 0.005


@pytest.mark.parametrize("code", [False, True])
def test_track_step_equals_the_tracker_loop(code):
    """``TrackStep`` (the tracker's iteration as a fixed launch sequence, eager and replayed from a hipGraph) against
    ``Tracker.track_frame`` (autograd driver, itself held to the oracle by test_gpu_slam.py) from the same seed: the same
    keep-best loss (1e-5) and camera (2e-5; float atomics in the pose gradient are the only run-to-run difference)."""
    from dns_slam_amd.fused_step import TrackStep
    from dns_slam_amd.tracking import Tracker
    cfg, bound, cam, frames, dec, mapper = _setup(64, 2, n_pixels=400)
    cfg["tracking"]["n_pixels"] = 256
    cur = {"gt_color": frames["gt_color"][2], "gt_depth": frames["gt_depth"][2], "gt_label": frames["gt_label"][2]}
    c2w = frames["est_c2w"][2].clone()
    c2w[:3, 3] += torch.tensor([0.02, -0.01, 0.015], dtype=c2w.dtype)
    feats = None
    if code:
        feats = (torch.rand(256, 32 + 15, 32, generator=torch.Generator().manual_seed(6)) * 2 - 1).to(DEV)
    out = {}
    for mode in ("autograd", "eager", "graph"):
        tracker = Tracker(cfg, dec, bound, cam, device=DEV)
        tracker.border = 5
        tracker.static_shapes = True
        torch.manual_seed(3)
        torch.cuda.manual_seed(3)
        if mode == "autograd":
            cam7, best = tracker.track_frame(cur, c2w, n_iters=25, features=feats, fused=True, graph=False)
        else:
            with tracker.frozen_scene():
                ts = TrackStep(tracker, cur, c2w, features=feats)
                cam7, best = ts.run(25, graph=(mode == "graph"))
        torch.cuda.synchronize()
        out[mode] = (cam7.detach().cpu().clone(), float(best))
    for mode in ("eager", "graph"):
        assert abs(out[mode][1] - out["autograd"][1]) <= 1e-5 * abs(out["autograd"][1]), (mode, out[mode][1], out["autograd"][1])
        assert float((out[mode][0] - out["autograd"][0]).abs().max()) <= 2e-5, (mode, out[mode][0], out["autograd"][0])
    assert float((out["autograd"][0][4:] - c2w[:3, 3]).abs().max()) > 0        # the pose did move


@pytest.mark.parametrize("tracker", [0, 1])
@pytest.mark.parametrize("N", [300, 4096, 5000])
def test_loss_rays_single_launch_equals_the_three_calls(tracker, N):
    """dns_loss_rays (ABI v10: the point pass's partial sums + ONE single-workgroup kernel for the rays' sums, the finalize and the
    rays' backward) against dns_loss_sums -> dns_loss_finalize -> dns_loss_bwd(rays): the same per-ray arithmetic (shared device
    functions), the sums differ only in the order of the fp32 adds (fixed-order tree instead of float atomics): 16 sums and
    16 outputs to 2e-6, gradients to 2e-6 of their scale (they carry the coefficients).  Mapper (with the point terms) and
    tracker mode, some rays invalid, N below / at / above a multiple of the workgroup."""
    lib, check, ptr, stream_ptr = _lib()
    from ctypes import c_float
    from dns_slam_amd import ops
    g = torch.Generator().manual_seed(20 + N + tracker)
    S, Cn, L = (1, 8, 1) if tracker else (24, 8, 33)
    P = N * S
    pc, pd = torch.rand(N, 3, generator=g).to(DEV), (torch.rand(N, generator=g) * 3).to(DEV)
    pv, ps = (torch.rand(N, generator=g) + 0.01).to(DEV), torch.randn(N, Cn, generator=g).to(DEV)
    gc, gd = torch.rand(N, 3, generator=g).to(DEV), (torch.rand(N, generator=g) * 3).to(DEV)
    gd[::7] = 0.0
    lab = torch.randint(0, Cn, (N,), generator=g).to(DEV)
    valid = (torch.rand(N, generator=g) > 0.1).to(torch.uint8).to(DEV)
    fine, coarse = torch.randn(P, L, generator=g).to(DEV), torch.randn(P, L, generator=g).to(DEV)
    z = torch.sort(torch.rand(N, S, generator=g) * 3 + 0.1, dim=1).values.to(DEV)
    one = torch.full((1,), 0.7, device=DEV)
    lam = (c_float * 8)(5.0, 5.0, 0.1, 0.0 if tracker else 10.0, 0.0 if tracker else 10.0, 0.0 if tracker else 10.0, 0.2, 0.05)
    f = lambda *s_: torch.full(s_, float("nan"), device=DEV)
    pts = (None, None, None) if tracker else (ptr(fine), ptr(coarse), ptr(z))
    var = ptr(pv) if tracker else None
    res = []
    for fused in (False, True):
        sums, out = f(ops.LOSS_SUMS_FLOATS), f(16)
        dcol, ddep, dvar, dsem = f(N, 3), f(N), f(N), f(N, Cn)
        dv = ptr(dvar) if tracker else None
        if fused:
            check(lib.dns_loss_rays(lam, N, S, Cn, L, tracker, ptr(pc), ptr(pd), var, ptr(ps), ptr(gc), ptr(gd), ptr(lab), ptr(valid), *pts,
                                    ptr(sums), ptr(out), ptr(one), ptr(dcol), ptr(ddep), dv, ptr(dsem), stream_ptr()), "loss_rays")
        else:
            check(lib.dns_loss_sums(lam, N, S, Cn, L, tracker, ptr(pc), ptr(pd), var, ptr(ps), ptr(gc), ptr(gd), ptr(lab), ptr(valid), *pts,
                                    ptr(sums), stream_ptr()), "loss_sums")
            check(lib.dns_loss_finalize(lam, N, S, Cn, L, tracker, ptr(sums), ptr(out), stream_ptr()), "loss_finalize")
            check(lib.dns_loss_bwd(lam, N, S, Cn, L, tracker, ptr(out), ptr(one), ptr(pc), ptr(pd), var, ptr(ps), ptr(gc), ptr(gd), ptr(lab),
                                   ptr(valid), *pts, ptr(dcol), ptr(ddep), dv, ptr(dsem), None, None, 0, stream_ptr()), "loss_bwd")
        torch.cuda.synchronize()
        used = [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13]
        res.append((sums[:10].cpu(), out[used].cpu(), dcol.cpu(), ddep.cpu(), (dvar if tracker else ddep).cpu(), dsem.cpu()))
    # dns_loss_finalize_bwd: the finalize inside the rays' backward kernel, from the SAME sums -> bit-identical to the two launches
    sums, out = f(ops.LOSS_SUMS_FLOATS), f(16)
    dcol, ddep, dvar, dsem = f(N, 3), f(N), f(N), f(N, Cn)
    dv = ptr(dvar) if tracker else None
    check(lib.dns_loss_sums(lam, N, S, Cn, L, tracker, ptr(pc), ptr(pd), var, ptr(ps), ptr(gc), ptr(gd), ptr(lab), ptr(valid), *pts,
                            ptr(sums), stream_ptr()), "loss_sums")
    ref_out, ref = f(16), [f(N, 3), f(N), f(N), f(N, Cn)]
    check(lib.dns_loss_finalize(lam, N, S, Cn, L, tracker, ptr(sums), ptr(ref_out), stream_ptr()), "loss_finalize")
    check(lib.dns_loss_bwd(lam, N, S, Cn, L, tracker, ptr(ref_out), ptr(one), ptr(pc), ptr(pd), var, ptr(ps), ptr(gc), ptr(gd), ptr(lab),
                           ptr(valid), *pts, ptr(ref[0]), ptr(ref[1]), ptr(ref[2]) if tracker else None, ptr(ref[3]), None, None, 0,
                           stream_ptr()), "loss_bwd")
    check(lib.dns_loss_finalize_bwd(lam, N, S, Cn, L, tracker, ptr(sums), ptr(out), ptr(one), ptr(pc), ptr(pd), var, ptr(ps), ptr(gc),
                                    ptr(gd), ptr(lab), ptr(valid), ptr(dcol), ptr(ddep), dv, ptr(dsem), stream_ptr()), "loss_finalize_bwd")
    torch.cuda.synchronize()
    used_t = torch.tensor([0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13])
    assert torch.equal(out.cpu()[used_t], ref_out.cpu()[used_t])
    assert torch.equal(dcol, ref[0]) and torch.equal(ddep, ref[1]) and torch.equal(dsem, ref[3])
    assert not tracker or torch.equal(dvar, ref[2])
    for a, b, name in zip(res[1], res[0], ("sums", "out", "d colour", "d depth", "d var", "d logits")):
        assert torch.isfinite(b).all(), name
        assert_close(a, b, rtol=2e-6, what=f"dns_loss_rays ({'tracker' if tracker else 'mapper'}, N={N}): {name}", elementwise=False)
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max()) + 1e-12, name


def test_track_step_is_reused_across_frames():
    """``Tracker.track_frame(use_track_step)`` keeps ONE TrackStep (buffers, prepared-weight storage, captured graph) and resets it
    per frame: the second frame of a cached, graph-replaying tracker must give what a FRESH tracker gives for that frame from
    the same generator state -- other images, other initial pose, and scene weights that changed in between (the prepared
    operand images must be rebuilt) --, and the graph must not have been captured twice."""
    from dns_slam_amd.tracking import Tracker
    cfg, bound, cam, frames, dec, mapper = _setup(64, 2, n_pixels=400)
    cfg["tracking"]["n_pixels"] = 256
    curs = [{"gt_color": frames["gt_color"][k], "gt_depth": frames["gt_depth"][k], "gt_label": frames["gt_label"][k]} for k in (1, 2)]
    c2ws = []
    for k in (1, 2):
        c = frames["est_c2w"][k].clone()
        c[:3, 3] += torch.tensor([0.02, -0.01, 0.015], dtype=c.dtype) * k
        c2ws.append(c)

    def perturb():
        with torch.no_grad():
            dec.coarse_fn.decoder.params.mul_(1.01)

    cached = Tracker(cfg, dec, bound, cam, device=DEV)
    cached.border, cached.use_track_step = 5, True
    torch.manual_seed(3)
    torch.cuda.manual_seed(3)
    cached.track_frame(curs[0], c2ws[0], n_iters=10, graph=True)
    first = cached.last_track_step
    g_first = first.g
    perturb()
    torch.cuda.manual_seed(7)
    cam_c, best_c = cached.track_frame(curs[1], c2ws[1], n_iters=20, graph=True)
    assert cached.last_track_step is first and first.g is g_first
    fresh = Tracker(cfg, dec, bound, cam, device=DEV)
    fresh.border, fresh.use_track_step = 5, True
    torch.cuda.manual_seed(7)
    cam_f, best_f = fresh.track_frame(curs[1], c2ws[1], n_iters=20, graph=True)
    torch.cuda.synchronize()
    assert abs(float(best_c) - float(best_f)) <= 1e-5 * abs(float(best_f)), (float(best_c), float(best_f))
    assert float((cam_c - cam_f).abs().max()) <= 2e-5
    # a different signature (another ray count) builds a new TrackStep
    cached.n_pixels = 128
    cached.track_frame(curs[1], c2ws[1], n_iters=2, graph=False)
    assert cached.last_track_step is not first


def test_tracker_glue_kernels():
    lib, check, ptr, stream_ptr = _lib()
    g = torch.Generator().manual_seed(8)
    N = 300
    d = torch.rand(N, generator=g) * 0.03
    ins = (torch.rand(N, generator=g) < 0.7).to(torch.uint8)
    valid = torch.empty(N, dtype=torch.uint8, device=DEV)
    d_d, ins_d = d.to(DEV), ins.to(DEV)
    check(lib.dns_track_mask(ptr(d_d), ptr(ins_d), N, 0.01, ptr(valid), stream_ptr()), "dns_track_mask")
    assert torch.equal(valid.cpu().bool(), (d > 0.01) & ins.bool())              # slams/tracking.py:171-172
    q, t = torch.tensor([[0.9, 0.1, 0.2, 0.3]], device=DEV), torch.tensor([[1.0, 2.0, 3.0]], device=DEV)
    best_loss, best_cam = torch.tensor([5.0], device=DEV), torch.zeros(7, device=DEV)
    for loss, want_l, want_q0 in ((7.0, 5.0, 0.0), (3.0, 3.0, 0.9), (float("nan"), 3.0, 0.9)):
        l = torch.tensor([loss], device=DEV)
        check(lib.dns_keep_best(ptr(l), ptr(q), ptr(t), ptr(best_loss), ptr(best_cam), stream_ptr()), "dns_keep_best")
        assert float(best_loss) == want_l and abs(float(best_cam[0]) - want_q0) < 1e-7
    assert torch.equal(best_cam.cpu(), torch.tensor([0.9, 0.1, 0.2, 0.3, 1.0, 2.0, 3.0]))
    tt = torch.rand(15, generator=g).to(DEV)
    keep = tt.clone()
    check(lib.dns_force_half(ptr(tt), 15, 8, stream_ptr()), "dns_force_half")        # utils/common.py:572-574
    keep[8] = 0.5
    assert torch.equal(tt, keep)
    tt[3] = 0.5
    tt[8] = 0.25
    keep = tt.clone()
    check(lib.dns_force_half(ptr(tt), 15, 8, stream_ptr()), "dns_force_half")        # a draw already equals 0.5: untouched
    assert torch.equal(tt, keep)


# ------------------------------------------------------------------------------------------------ the glue kernels (csrc/step.hip)
def _lib():
    from dns_slam_amd import ops
    from dns_slam_amd._lib import check, ptr, stream_ptr
    return ops.lib, check, ptr, stream_ptr


@pytest.mark.parametrize("tiled", [1, 0])
def test_class_slots_equals_the_reference_label_tiling(tiled):
    lib, check, ptr, stream_ptr = _lib()
    g = torch.Generator().manual_seed(3)
    N, S = 97, 13
    labels = torch.randint(-2, 12, (N,), generator=g)
    lut = torch.full((10,), -1, dtype=torch.int64)
    lut[[0, 2, 3, 7]] = torch.tensor([0, 1, 2, 3])
    out = torch.empty(N * S, dtype=torch.int64, device=DEV)
    lab_d, lut_d = labels.to(DEV), lut.to(DEV)
    check(lib.dns_class_slots(ptr(lab_d), N, S, tiled, ptr(lut_d), 10, ptr(out), stream_ptr()), "dns_class_slots")
    classes = labels.repeat(1, S).flatten(0, 1) if tiled else labels.repeat_interleave(S)      # slams/mapping.py:613 / per ray
    want = torch.where((classes >= 0) & (classes < 10), lut[classes.clamp(0, 9)], torch.full_like(classes, -1))
    assert torch.equal(out.cpu(), want)


@pytest.mark.parametrize("with_code", [True, False])
def test_feature_block_rgb_sigmoid_and_their_backward(with_code):
    lib, check, ptr, stream_ptr = _lib()
    g = torch.Generator().manual_seed(4)
    N, S, H, Cc = 50, 9, 32, 32
    P = N * S
    fine = torch.randn(P, H + 1, generator=g)
    code = torch.rand(N, S, Cc, generator=g) * 2 - 1
    d = torch.rand(N, generator=g) * 3
    d[::7] = 0.0
    z = d[:, None] * (0.8 + 0.4 * torch.rand(N, S, generator=g)) + 0.01
    raw = torch.randn(P, 4, generator=g)
    feat_d, raw_d = torch.empty(P, H + Cc, device=DEV), raw.to(DEV)
    fine_d, code_d, z_d, d_d = fine.to(DEV), code.to(DEV), z.to(DEV), d.to(DEV)
    check(lib.dns_feature_block(ptr(fine_d), H + 1, H, ptr(code_d) if with_code else None, Cc, ptr(z_d), ptr(d_d), N, S, ptr(feat_d),
                                H + Cc, ptr(raw_d), stream_ptr()), "dns_feature_block")
    check(lib.dns_rgb_sigmoid(ptr(raw_d), P, stream_ptr()), "dns_rgb_sigmoid")
    dd = d[:, None]
    trunc = (1.0 - (z < dd * 0.95).float()) * (1.0 - (z > dd * 1.05).float()) * (dd > 0.0).float()     # slams/mapping.py:553-556
    assert 0.1 < float(trunc.mean()) < 0.9
    want_code = (code * trunc[..., None]).reshape(P, Cc) if with_code else torch.zeros(P, Cc)
    assert torch.equal(feat_d.cpu(), torch.cat((fine[:, 1:], want_code), -1))
    want_raw = torch.cat((torch.sigmoid(raw[:, :3]), fine[:, 0:1]), -1)
    assert_close(raw_d.cpu(), want_raw, rtol=1e-6, what="rgb sigmoid | occupancy")
    # backward: d_col = d_raw * s (1 - s) on the colour columns, d occupancy added into a strided column
    d_raw = torch.randn(P, 4, generator=g)
    wide = torch.randn(P, 7, generator=g)
    d_raw_d, wide_d, d_col = d_raw.to(DEV), wide.to(DEV), torch.empty(P, 4, device=DEV)
    from ctypes import c_void_p
    for acc in (1, 0):
        w = wide_d.clone()
        check(lib.dns_raw_bwd(ptr(d_raw_d), ptr(raw_d), P, ptr(d_col), c_void_p(w.data_ptr() + 4 * 3), 7, acc, stream_ptr()), "dns_raw_bwd")
        s = want_raw[:, :3]
        assert_close(d_col.cpu()[:, :3], d_raw[:, :3] * s * (1 - s), rtol=1e-6, what="sigmoid backward")
        assert torch.equal(d_col.cpu()[:, 3], torch.zeros(P))
        want_w = wide.clone()
        want_w[:, 3] = wide[:, 3] + d_raw[:, 3] if acc else d_raw[:, 3]
        assert torch.equal(w.cpu(), want_w)


def test_lattice_points_equal_the_float64_affine_map():
    lib, check, ptr, stream_ptr = _lib()
    from ctypes import c_double
    cfg, bound, cam, frames, dec, mapper = _setup()
    sp = cfg["training"]["smooth_pts"]
    mapper._ensure_lattice(sp, 0.1, 0.05)
    _, c_vox, c_off, c_mar = mapper._lattice_consts
    r = torch.rand(6, device=DEV)
    r64 = r.to(torch.float64)
    b = torch.addcmul(torch.addcmul(c_mar, r64[:3], c_off), r64[3:], c_vox)                  # Mapper.smoothness, static_shapes
    want = torch.addcmul(b, mapper._lattice, c_vox).reshape(-1, 3).float()
    n = sp - 1
    got = torch.empty(n ** 3, 3, device=DEV)
    c9 = (c_double * 9)(*[float(v) for t in (c_vox, c_off, c_mar) for v in t.cpu().tolist()])
    check(lib.dns_lattice_points(ptr(r), c9, n, None, 0, ptr(got), stream_ptr()), "dns_lattice_points")
    # float64 arithmetic rounded to float32 once: at most the last bit where torch contracts a multiply-add
    assert float((got - want).abs().max()) <= 6e-8 * float(want.abs().max())
    assert float((got != want).float().mean()) < 1e-3
    order = torch.randperm(n ** 3, device=DEV).to(torch.int32)            # any element order: row m = element order[m]
    got2 = torch.empty(n ** 3, 3, device=DEV)
    check(lib.dns_lattice_points(ptr(r), c9, n, ptr(order), 0, ptr(got2), stream_ptr()), "dns_lattice_points")
    assert torch.equal(got2, got[order.long()])
    part = order[100:100 + 777].contiguous()                              # a sub-list (a rank's slab of the lattice): count rows
    got3 = torch.full((777 + 5, 3), -7.0, device=DEV)
    check(lib.dns_lattice_points(ptr(r), c9, n, ptr(part), 777, ptr(got3), stream_ptr()), "dns_lattice_points")
    assert torch.equal(got3[:777], got[part.long()]) and bool((got3[777:] == -7.0).all())


def test_draw_finish_equals_the_mapper_draw_arithmetic():
    """dns_draw_finish on given random numbers == Mapper.draw_pixels' index arithmetic (select_uv + select_by_class,
    utils/common.py:274,313-328), the labels raygen would read and the per-frame depth maxima."""
    lib, check, ptr, stream_ptr = _lib()
    cfg, bound, cam, frames, dec, mapper = _setup()
    prep = mapper.prepare_frames(frames)
    K, n1, n2, HW = 4, prep["n1"], prep["n2"], prep["HW"]
    npf = n1 + n2
    torch.manual_seed(21)
    i1 = torch.randint(HW, (K, n1), device=DEV)
    u = torch.rand(K, n2, device=DEV, dtype=torch.float64)
    j = torch.minimum((u * prep["counts_f64"]).to(torch.int64), prep["counts_m1"])
    want_pix = torch.cat((i1, prep["sorted_flat"][prep["starts_flat"] + j]), 1).reshape(-1)
    pix = torch.empty(K * npf, device=DEV, dtype=torch.int64)
    labels = torch.empty(K * npf, device=DEV, dtype=torch.int64)
    dmax = torch.empty(K, device=DEV, dtype=torch.int32)
    check(lib.dns_draw_finish(ptr(i1), ptr(u), ptr(prep["counts_f64"]), ptr(prep["counts_m1"]), ptr(prep["starts_flat"]),
                              ptr(prep["sorted_flat"]), ptr(prep["depth"]), ptr(prep["label"]), K, n1, n2, HW, ptr(pix), ptr(labels),
                              ptr(dmax), stream_ptr()), "dns_draw_finish")
    assert torch.equal(pix, want_pix)
    p2 = want_pix.reshape(K, npf)
    assert torch.equal(labels, torch.gather(prep["label"].reshape(K, -1), 1, p2).reshape(-1).long())
    assert torch.equal(dmax.view(torch.float32), torch.gather(prep["depth"].reshape(K, -1), 1, p2).amax(dim=1).clamp_min(0.0))


def test_draw_finish_equals_the_oracle_class_balanced_pick():
    """a2 on the GPU against the ORACLE: dns_draw_finish's class-balanced half == oracle.render_math.class_balanced_indices
    (pinned to the imported reference's select_by_class by tests/golden/get_samples_by_class.npz, utils/common.py:307-338)
    when the oracle's per-class ``draw(k, m)`` takes its integers from the same uniforms u the kernel is given
    (floor(u k), clamped to k - 1): same classes in ascending order, same remainder rule for the first class (:318-321), same
    ascending pixel list per class, a one-pixel class repeated (:324-325).  The uniform half is select_uv's plain randint (:274)."""
    lib, check, ptr, stream_ptr = _lib()
    from oracle import render_math as rm
    cfg, bound, cam, frames, dec, mapper = _setup()
    # a class with exactly ONE pixel in frame 1 (the k == 1 branch) -- written before the class tables are built
    frames["gt_label"][1] = frames["gt_label"][1].clone()
    frames["gt_label"][1][3, 5] = 37.0
    prep = mapper.prepare_frames(frames)
    K, n1, n2, HW = 4, prep["n1"], prep["n2"], prep["HW"]
    npf = n1 + n2
    torch.manual_seed(77)
    i1 = torch.randint(HW, (K, n1), device=DEV)
    u = torch.rand(K, n2, device=DEV, dtype=torch.float64)
    pix = torch.empty(K * npf, device=DEV, dtype=torch.int64)
    labels = torch.empty(K * npf, device=DEV, dtype=torch.int64)
    dmax = torch.empty(K, device=DEV, dtype=torch.int32)
    check(lib.dns_draw_finish(ptr(i1), ptr(u), ptr(prep["counts_f64"]), ptr(prep["counts_m1"]), ptr(prep["starts_flat"]),
                              ptr(prep["sorted_flat"]), ptr(prep["depth"]), ptr(prep["label"]), K, n1, n2, HW, ptr(pix), ptr(labels),
                              ptr(dmax), stream_ptr()), "dns_draw_finish")
    got = pix.reshape(K, npf).cpu()
    u_h = u.cpu()
    saw_single = False
    for f in range(K):
        lab = frames["gt_label"][f].cpu().float()
        counts = torch.unique(lab.reshape(-1), return_counts=True)[1]
        n_class = counts.numel()
        n_k = n2 // n_class
        m_of = [n2 - n_k * (n_class - 1) if c == 0 else n_k for c in range(n_class)]
        first = [sum(m_of[:c]) for c in range(n_class)]          # the kernel's uniforms are laid out class after class
        # the oracle (like the reference) draws nothing for a one-pixel class: its draw() calls are the other classes, ascending
        calls = iter([c for c in range(n_class) if int(counts[c]) != 1])
        saw_single = saw_single or any(int(counts[c]) == 1 for c in range(n_class))

        def draw(k, m):
            c = next(calls)
            assert m == m_of[c] and k == int(counts[c])
            uu = u_h[f, first[c]:first[c] + m]
            return torch.minimum((uu * float(k)).to(torch.int64), torch.tensor(k - 1))

        want = rm.class_balanced_indices(lab, n2, draw=draw)
        assert torch.equal(got[f, n1:], want), f
        assert torch.equal(got[f, :n1], i1[f].cpu())
    assert saw_single


def test_composite_rgb_logits_and_point_loss_backward_with_occupancy():
    """ABI v9 glue fusions of the mapping step: dns_composite_fwd_ex / _bwd_ex with DNS_COMPOSITE_RGB_LOGITS == dns_rgb_sigmoid +
    dns_composite_fwd / dns_composite_bwd + dns_raw_bwd (the colour network's sigmoid, models/decoder.py:124, folded into the
    compositing); dns_loss_bwd (rays only) + dns_loss_bwd_points(d_occ) == dns_loss_bwd followed by the strided += of d occupancy."""
    lib, check, ptr, stream_ptr = _lib()
    from ctypes import c_float, c_void_p
    from dns_slam_amd import ops
    g = torch.Generator().manual_seed(12)
    N, S, Cn, L = 300, 47, 8, 33
    P = N * S
    raw = torch.randn(P, 4, generator=g).to(DEV)
    z = torch.sort(torch.rand(N, S, generator=g) * 3 + 0.1, dim=1).values.to(DEV)
    logit = torch.randn(P, Cn, generator=g).to(DEV)
    f = lambda *s: torch.empty(*s, device=DEV)
    outs = []
    for fused in (False, True):
        r = raw.clone()
        depth, var, rgb, w, sem = f(N), f(N), f(N, 3), f(N, S), f(N, Cn)
        if fused:
            check(lib.dns_composite_fwd_ex(ptr(r), ptr(z), ptr(logit), N, S, Cn, ptr(depth), ptr(var), ptr(rgb), ptr(w), ptr(sem), 1,
                                           stream_ptr()), "fwd_ex")
        else:
            check(lib.dns_rgb_sigmoid(ptr(r), P, stream_ptr()), "sig")
            check(lib.dns_composite_fwd(ptr(r), ptr(z), ptr(logit), N, S, Cn, ptr(depth), ptr(var), ptr(rgb), ptr(w), ptr(sem),
                                        stream_ptr()), "fwd")
        gd, gc, gs = torch.randn(N, generator=torch.Generator().manual_seed(1)).to(DEV), torch.randn(N, 3, generator=torch.Generator().manual_seed(2)).to(DEV), \
            torch.randn(N, Cn, generator=torch.Generator().manual_seed(3)).to(DEV)
        d_raw, d_logit = f(P, 4), f(P, Cn)
        wide = torch.full((P, 7), 0.5, device=DEV)
        if fused:
            check(lib.dns_composite_bwd_ex(ptr(r), ptr(z), ptr(logit), N, S, Cn, ptr(gd), None, ptr(gc), None, ptr(gs), ptr(d_raw), ptr(d_logit),
                                           1, stream_ptr()), "bwd_ex")
            d_col = d_raw.clone()
            d_col[:, 3] = 0.0
            wide[:, 3] += d_raw[:, 3]
        else:
            check(lib.dns_composite_bwd(ptr(r), ptr(z), ptr(logit), N, S, Cn, ptr(gd), None, ptr(gc), None, ptr(gs), ptr(d_raw), ptr(d_logit),
                                        stream_ptr()), "bwd")
            d_col = f(P, 4)
            check(lib.dns_raw_bwd(ptr(d_raw), ptr(r), P, ptr(d_col), c_void_p(wide.data_ptr() + 12), 7, 1, stream_ptr()), "raw_bwd")
        outs.append([depth, var, rgb, w, sem, d_col, d_logit, wide])
    torch.cuda.synchronize()
    for a, b, name in zip(outs[1], outs[0], ("depth", "var", "rgb", "weights", "sem", "d colour logits", "d logits", "d occupancy")):
        assert_close(a.cpu(), b.cpu(), rtol=1e-6, what=f"rgb-logit compositing: {name}")
    # ---- point losses with the occupancy gradient added
    fine, coarse = torch.randn(P, L, generator=g).to(DEV), torch.randn(P, L, generator=g).to(DEV)
    gt_depth = (torch.rand(N, generator=g) * 3).to(DEV)
    valid = (torch.rand(N, generator=g) > 0.1).to(torch.uint8).to(DEV)
    out16 = torch.rand(16, generator=g).to(DEV)
    one = torch.ones(1, device=DEV)
    lam = (c_float * 8)(5.0, 5.0, 0.1, 10.0, 10.0, 10.0, 0.2, 0.05)
    d_occ = torch.randn(P, 4, generator=g).to(DEV)
    ldf = 4 + 64
    res = []
    for fused in (False, True):
        dfx, dco = torch.zeros(P, ldf, device=DEV), f(P, L)
        dst = c_void_p(dfx.data_ptr() + 12)
        dcol, ddep, dsem = f(N, 3), f(N), f(N, Cn)
        pc, pd, ps = torch.rand(N, 3, generator=torch.Generator().manual_seed(5)).to(DEV), torch.rand(N, generator=torch.Generator().manual_seed(6)).to(DEV), \
            torch.randn(N, Cn, generator=torch.Generator().manual_seed(7)).to(DEV)
        gcol, lab = torch.rand(N, 3, generator=torch.Generator().manual_seed(8)).to(DEV), torch.randint(0, Cn, (N,), generator=torch.Generator().manual_seed(9)).to(DEV)
        args = (lam, N, S, Cn, L, 0, ptr(out16), ptr(one), ptr(pc), ptr(pd), None, ptr(ps), ptr(gcol), ptr(gt_depth), ptr(lab), ptr(valid),
                ptr(fine), ptr(coarse), ptr(z), ptr(dcol), ptr(ddep), None, ptr(dsem))
        if fused:
            check(lib.dns_loss_bwd(*args, None, None, 0, stream_ptr()), "loss_bwd rays")
            check(lib.dns_loss_bwd_points(lam, N, S, Cn, L, ptr(out16), ptr(one), ptr(gt_depth), ptr(valid), ptr(fine), ptr(coarse), ptr(z),
                                          dst, ptr(dco), ldf, c_void_p(d_occ.data_ptr() + 12), 4, stream_ptr()), "loss_bwd_points")
        else:
            check(lib.dns_loss_bwd(*args, dst, ptr(dco), ldf, stream_ptr()), "loss_bwd")
            dfx[:, 3] += d_occ[:, 3]
        res.append((dfx, dco, dcol, ddep, dsem))
    torch.cuda.synchronize()
    for (a, b), name in zip(zip(res[1], res[0]), ("d fine (wide rows)", "d coarse", "d colour", "d depth", "d sem")):
        if name.startswith("d fine"):       # the occupancy term joins inside one fused multiply-add instead of a later rounded add
            assert_close(a.cpu(), b.cpu(), rtol=1e-6, what=f"point losses + d occupancy: {name}")
        else:
            assert torch.equal(a, b), name


@pytest.mark.parametrize("code", [False, True])
@pytest.mark.parametrize("nn,nl,nu,ns", [(64, 2, 32, 15), (64, 2, 48, 16), (32, 1, 22, 10)])
def test_track_step_fused_kernel_equals_the_launch_sequence(nn, nl, nu, ns, code):
    """``TrackStep.run_fused`` (round 5: the tracker's iteration as ONE kernel + a pose kernel, csrc/track_fused.inc) against the
    28-launch ``TrackStep.step`` on the SAME draws (pixels, surface jitter with its forced mid sample, zero-depth jitter): the
    same per-ray and per-point arithmetic (shared device functions, the same MLP bodies); what differs is the order of the
    fp32 sums of the pose gradient and ONE division by the valid-ray count instead of one per ray.  15 free-running iterations:
    best loss to 1e-5, pose and best camera to 2e-5, the last iteration's loss terms to 2e-5 of themselves.  S = 47, 64 and 32
    (two and four rays per workgroup), 64 x 2 and 32 x 1 networks, with and without a per-sample 2-D code; eager and replayed
    from ONE captured iteration."""
    from dns_slam_amd.fused_step import TrackStep
    from dns_slam_amd.tracking import Tracker
    cfg, bound, cam, frames, dec, mapper = _setup(nn, nl, n_pixels=400)
    cfg["tracking"]["n_pixels"] = 250
    cfg["training"]["n_samples_ray"], cfg["training"]["n_surface_ray"] = nu, ns
    cur = {"gt_color": frames["gt_color"][2], "gt_depth": frames["gt_depth"][2], "gt_label": frames["gt_label"][2]}
    c2w = frames["est_c2w"][2].clone()
    c2w[:3, 3] += torch.tensor([0.02, -0.01, 0.015], dtype=c2w.dtype)
    n_it, N = 15, 250
    feats = (torch.rand(N, nu + ns, 32, generator=torch.Generator().manual_seed(6)) * 2 - 1).to(DEV) if code else None
    g = torch.Generator().manual_seed(9)

    def mk():
        tracker = Tracker(cfg, dec, bound, cam, device=DEV)
        tracker.border = 5
        tracker.static_shapes = True
        return tracker

    tracker = mk()
    H, W, b = tracker.H, tracker.W, tracker.border
    draws = (torch.randint((H - 2 * b) * (W - 2 * b), (n_it, N), generator=g), torch.rand(n_it, tracker.n_surface_ray, generator=g),
             torch.rand(n_it, tracker.n_surface_ray, generator=g))
    res = {}
    for mode in ("fused", "fused_graph"):
        tracker = mk()
        with tracker.frozen_scene():
            ts = TrackStep(tracker, cur, c2w, features=feats)
            assert ts.fused_supported()
            cam7, best = ts.run_fused(n_it, graph=(mode == "fused_graph"), draws=draws)
            torch.cuda.synchronize()
            t_surf_forced = ts._fb["keep"]["t_surf"].clone()
            res[mode] = (cam7.cpu().clone(), float(best), ts.Q.cpu().clone(), ts.T.cpu().clone(), ts.fused_out.cpu().clone())
    # the launch sequence on the same draws (the forced mid sample as dns_track_fused_begin left it)
    tracker = mk()
    with tracker.frozen_scene():
        ts = TrackStep(tracker, cur, c2w, features=feats)
        for k in range(n_it):
            ts.step(draws=(draws[0][k].to(DEV), t_surf_forced[k].contiguous(), draws[2][k].to(DEV)))
        torch.cuda.synchronize()
        ref = (ts.best_cam.cpu().clone(), float(ts.best_loss[0]), ts.Q.cpu().clone(), ts.T.cpu().clone(), ts.out.cpu().clone())
    for mode in ("fused", "fused_graph"):
        cam7, best, Q, T, out = res[mode]
        assert abs(best - ref[1]) <= 1e-5 * abs(ref[1]), (mode, best, ref[1])
        assert float((cam7 - ref[0]).abs().max()) <= 2e-5, (mode, cam7, ref[0])
        assert float((Q - ref[2]).abs().max()) <= 2e-5 and float((T - ref[3]).abs().max()) <= 2e-5, (mode, Q, ref[2], T, ref[3])
        # last iteration's terms: fused out = (p, d, l, total, n_valid); launch sequence out = (p, d, l, lt, fs, op, total, ...)
        for i, j in ((0, 0), (1, 1), (2, 2), (3, 6)):
            assert abs(float(out[i]) - float(ref[4][j])) <= 2e-5 * abs(float(ref[4][j])) + 1e-7, (mode, i, float(out[i]), float(ref[4][j]))
    assert float((ref[3].reshape(-1) - c2w[:3, 3].float()).abs().max()) > 0              # the pose did move


def test_map_step_morton_ordered_draws_are_the_same_draw():
    """mapper.morton_draws (label_layout 'per_ray' only, off by default): every frame's drawn pixel list in
    Morton order of (row, col) -- the same SET of pixels with their labels, so the step's losses equal those of the draw-order
    list up to the order of the sums; under the reference-tiled layout the option does nothing."""
    from dns_slam_amd.fused_step import MapStep
    out = {}
    for layout, morton in (("per_ray", False), ("per_ray", True), ("reference_tiled", True)):
        cfg, bound, cam, frames, dec, mapper = _setup(32, 1, layout=layout)
        mapper.static_shapes, mapper.is_BA, mapper.overlap_smooth, mapper.prefetch_draws = True, True, False, False
        mapper.morton_draws = morton
        _, ql, Tl = mapper.set_optimizer(frames, fused=True)
        ms = MapStep(mapper, frames, ql, Tl)
        assert ms.morton == (morton and layout == "per_ray")
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        ms.step()
        K, npf = ms.K, ms.npf
        pix = ms.cur.draws["pix"].view(K, npf).clone()
        out[(layout, morton)] = (pix, float(ms.losses()[0]), ms.gt_label.view(K, npf).clone())
    p0, l0, lab0 = out[("per_ray", False)]
    p1, l1, lab1 = out[("per_ray", True)]
    assert torch.equal(torch.sort(p0, 1)[0], torch.sort(p1, 1)[0]) and not torch.equal(p0, p1)
    W = mapper.W
    code = lambda p: sum((((p // W) >> b) & 1) << (2 * b + 1) | (((p % W) >> b) & 1) << (2 * b) for b in range(10))
    c1 = code(p1)
    assert bool((c1[:, 1:] >= c1[:, :-1]).all()), "not in Morton order"
    # the labels travel with their pixels
    o0, o1 = torch.argsort(p0, dim=1, stable=True), torch.argsort(p1, dim=1, stable=True)
    assert torch.equal(torch.gather(lab0, 1, o0), torch.gather(lab1, 1, o1))
    assert abs(l0 - l1) <= 1e-4 * abs(l0)
