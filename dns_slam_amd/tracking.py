"""Mirror of the Tracker optimise-step bodies (reference slams/tracking.py) on the gfx950 kernels.

    compute_*_loss        slams/tracking.py:85-96
    set_optimizer         :108-126
    get_target_samples    :128-186  (20-px border window; 2-D feature branch = an input)
    renderer              :188-214  (coarse-only)
    track_frame           the per-frame loop body :313-340 (keep-best-pose included)

The process loop (``run`` :229: data loading, mapper synchronisation, logging) is out of scope.
"""
from __future__ import annotations

import contextlib

import torch
import torch.nn.functional as F

from . import ops
from .common import feature_matching, get_quad_from_c2w


class Tracker:
    def __init__(self, cfg: dict, decoder, bound: torch.Tensor, cam: dict, device="cuda"):
        self.cfg = cfg
        self.device = device
        self.decoder = decoder
        self.bound = bound.to(torch.float64)
        self.H, self.W = cam["H"], cam["W"]
        self.fx, self.fy, self.cx, self.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
        tr, tk = cfg["training"], cfg["tracking"]
        self.lambda_p, self.lambda_d, self.lambda_l = tr["lambda_color"], tr["lambda_depth"], tr["lambda_label"]
        self.n_samples_ray, self.n_surface_ray = tr["n_samples_ray"], tr["n_surface_ray"]
        self.cam_lr, self.n_iters, self.n_pixels = tk["cam_lr"], tk["n_iters"], tk["n_pixels"]
        self.seperate_LR = cfg.get("seperate_LR", False)
        self.hidden_dim = decoder.hidden_dim
        self.pe_dim = decoder.pe_dim
        self.border = 20
        self.static_shapes = False        # True: device-side jitter draws, no host work per iteration (hipGraph-capturable)
        self.use_track_step = False       # True: track_frame runs fused_step.TrackStep (fixed launch sequence, device-side draws)
        self.t_uniform = torch.linspace(0.0, 1.0, steps=self.n_samples_ray, device=device) if self.n_samples_ray > 0 else None
        if str(device) != "cpu":
            from ._lib import ensure_init
            ensure_init()                         # dns_init() outside any capture (see Mapper.__init__)
            self._jitter_consts()                 # built on the device now: nothing is copied host -> device inside a capture

    def _jitter_consts(self):
        ns = self.n_surface_ray
        if getattr(self, "_force_mask", None) is None or self._force_mask.numel() != ns:
            self._force_mask = torch.arange(ns, device=self.device) == (ns // 2 + 1)
            self._half = torch.full((ns,), 0.5, device=self.device)

    # slams/tracking.py:85-96 -- masked means written as weighted sums (no boolean-mask gather / host sync)
    def compute_photometric_loss(self, gt_color, pred_color, mask):
        w = mask.to(gt_color.dtype)
        return (((gt_color - pred_color) ** 2) * w[:, None]).sum() / (w.sum() * gt_color.shape[1])

    def compute_depth_loss(self, gt_depth, pred_depth, pred_depth_var, mask):
        w = mask.to(gt_depth.dtype)
        return ((torch.abs(gt_depth - pred_depth) / torch.sqrt(pred_depth_var + 1e-10)) * w).sum() / w.sum()

    def compute_label_loss(self, gt_label, pred_logits, mask):
        w = mask.to(pred_logits.dtype)
        ce = F.cross_entropy(pred_logits, gt_label, reduction="none")
        return (ce * w).sum() / w.sum()

    # slams/tracking.py:108-126
    def set_optimizer(self, c2w, idx=None, fused=False):
        lr = self.cam_lr
        quad = get_quad_from_c2w(c2w).clone().detach().to(self.device).requires_grad_(True)
        T = c2w[:3, 3].clone().detach().to(self.device).float().requires_grad_(True)
        lrT = lr * 0.2 if self.seperate_LR else lr
        groups = [{"params": [T], "lr": lrT}, {"params": [quad], "lr": lr}]
        if fused:
            from .optim import FusedAdam
            return FusedAdam(groups), quad, T
        return torch.optim.Adam(groups), quad, T

    def draw_pixels(self):
        b = self.border
        return torch.randint((self.H - 2 * b) * (self.W - 2 * b), (self.n_pixels,), device=self.device)

    def draw_jitter(self):
        ns = self.n_surface_ray
        if self.static_shapes:
            self._jitter_consts()
            t = torch.rand(ns, device=self.device)
            t = torch.where(self._force_mask & ~(t == 0.5).any(), self._half, t)
            return t, torch.rand(ns, device=self.device)
        t = torch.rand(ns)
        if not torch.any(t == 0.5):
            t[ns // 2 + 1] = 0.5
        t0 = torch.rand(ns)
        return t.to(self.device), t0.to(self.device)

    def prepare_frame(self, cur_frames):
        f = lambda t: t.to(self.device).float().contiguous()[None]
        return {"color": f(cur_frames["gt_color"]), "depth": f(cur_frames["gt_depth"]), "label": f(cur_frames["gt_label"])}

    # slams/tracking.py:128-186
    def get_target_samples(self, cur_frames, refer_frames=None, features=None, prep=None, pix_idx=None, jitter=None):
        if prep is None:
            prep = self.prepare_frame(cur_frames)
        if pix_idx is None:
            pix_idx = self.draw_pixels()
        if jitter is None:
            jitter = self.draw_jitter()
        b = self.border
        rays_o, rays_d, pts, gt_color, gt_depth, gt_label, inside, z = ops.raygen_sample(
            cur_frames["est_quad"][None], cur_frames["est_T"][None], pix_idx, prep["color"], prep["depth"], prep["label"],
            (self.fx, self.fy, self.cx, self.cy), self.bound, (b, self.H - b, b, self.W - b), pix_idx.numel(),
            self.t_uniform, jitter[0], jitter[1])
        N, S = z.shape
        if features is None:
            code = torch.zeros(N, S, self.hidden_dim, device=self.device)
        else:
            if features.dim() == 5:
                # stem maps [1, n_refer, C, h, w] + refer_frames['est_w2c'] [n_refer,4,4]: slams/tracking.py:162-165
                K = torch.tensor([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]])
                w2c = refer_frames["est_w2c"].clone().detach()
                features = feature_matching(self.H, self.W, K, pts.flatten(0, 1), w2c, features[0],
                                            self.decoder.merge).reshape(N, S, -1)
            d = gt_depth[:, None]
            trunc = (1.0 - (z < d * 0.95).float()) * (1.0 - (z > d * 1.05).float()) * (d > 0.0).float()
            code = features * trunc[..., None]
        mask = (gt_depth > 0.01) & inside.bool()            # :171-172; kept on device (reference: .cpu().numpy())
        return {"gt_color": gt_color, "gt_depth": gt_depth, "gt_label": gt_label, "rays_o": rays_o, "rays_d": rays_d,
                "pts": pts, "z_vals": z, "mask": mask, "features": code}

    # slams/tracking.py:188-214
    def renderer(self, samples):
        pts = samples["pts"]
        z_vals = samples["z_vals"]
        n_pts, n_samples = z_vals.shape
        pixel_pts = samples["features"].flatten(0, 1)
        buf = self.decoder.pe_fn.forward_world(pts.flatten(0, 1), self.bound)
        pe, grid_pts = buf[:, :self.pe_dim], buf[:, self.pe_dim:]
        latents = self.decoder.coarse_fn(pe, features=grid_pts)
        color_pts, logits_pts = self.decoder.out_fn(pe, torch.cat((latents[:, 1:], pixel_pts), -1))
        values_pts = torch.cat((color_pts, latents[:, 0:1]), -1).reshape(n_pts, n_samples, -1)
        logits_pts = logits_pts.reshape(n_pts, n_samples, -1)
        pred_depth, pred_depth_var, pred_color, weights, pred_logits = ops.composite(values_pts, z_vals, logits_pts)
        return pred_color, pred_depth, pred_depth_var, pred_logits

    @contextlib.contextmanager
    def frozen_scene(self):
        """Tracking optimises the pose only (slams/tracking.py:120-124).  The reference still lets autograd compute the
        gradients of the (deep-copied) scene parameters and throws them away; here the scene is frozen for the duration of
        the loop, which skips every weight-gradient GEMM and the hash-grid scatter without changing the pose gradients."""
        params = [p for p in self.decoder.parameters() if p.requires_grad]
        for p in params:
            p.requires_grad_(False)
        try:
            yield
        finally:
            for p in params:
                p.requires_grad_(True)

    # slams/tracking.py:313-340
    def track_frame(self, cur_frames, est_c2w, n_iters=None, features=None, fused=False, graph=False, graph_warmup=0,
                    refer_frames=None):
        """Optimise (quat, T) of one frame against the frozen scene; returns the best-loss camera tensor [7].
        ``graph=True`` captures ONE iteration (sampling, render, losses, backward, fused Adam, keep-best) into a hipGraph
        and replays it n_iters times: tracking is 30-50 tiny latency-bound iterations per frame."""
        n_iters = self.n_iters if n_iters is None else n_iters
        if getattr(self, "use_track_step", False) and (features is None or features.dim() == 3 or refer_frames is not None):
            # the same loop as a fixed launch sequence over preallocated buffers (no autograd graph; draws on the device
            # generator like static_shapes); graph=True replays one captured iteration.  `fused` / `graph_warmup` do not apply
            # (Adam is the library's own kernel, a capture needs no warm-up); there is no torch optimizer object afterwards
            # ONE TrackStep per tracker and signature (shapes, constants, parameter addresses): its ~30 buffers, the prepared
            # weight images' storage and the captured graph are reused from frame to frame; only the frame's data, the pose and
            # the optimiser state are reset (a fresh TrackStep per frame cost a capture and ~30 allocations per frame)
            from .fused_step import TrackStep
            key = TrackStep.signature(self, features, refer_frames)
            ts = getattr(self, "_track_step", None)
            if ts is not None and getattr(self, "_track_step_key", None) == key:
                ts.reset(cur_frames, est_c2w, features=features, refer_frames=refer_frames)
            else:
                ts = TrackStep(self, cur_frames, est_c2w, features=features, refer_frames=refer_frames)
                self._track_step, self._track_step_key = ts, key
            # round 5: the whole iteration as ONE kernel + a pose kernel (csrc/track_fused.inc) wherever the shape allows it
            # (S <= 64, 64 x 2 / 32 x 1 networks, no in-loop stem branch); use_fused_kernel = False keeps the launch sequence
            if getattr(self, "use_fused_kernel", True) and ts.fused_supported():
                cam, best = ts.run_fused(n_iters, graph=graph)
            else:
                cam, best = ts.run(n_iters, graph=graph)
            cam, best = cam.clone(), best.clone()          # the TrackStep's own buffers are overwritten by the next frame
            self.last_track_step = ts
            self.last_optimizer = None
            return cam, best
        with self.frozen_scene():
            if graph:
                return self._track_frame_graphed(cur_frames, est_c2w, n_iters, features, graph_warmup, refer_frames)
            return self._track_frame_eager(cur_frames, est_c2w, n_iters, features, fused, refer_frames)

    def _track_frame_eager(self, cur_frames, est_c2w, n_iters, features, fused, refer_frames=None):
        optimizer, quad, T = self.set_optimizer(est_c2w, fused=fused)
        frames = dict(cur_frames)
        frames["est_quad"], frames["est_T"] = quad, T
        prep = self.prepare_frame(cur_frames)
        best_loss = torch.full((), float("inf"), device=self.device)
        best_cam = torch.cat((quad, T), 0).detach().clone()
        for _ in range(n_iters):
            optimizer.zero_grad()
            samples = self.get_target_samples(frames, refer_frames=refer_frames, features=features, prep=prep)
            pred_color, pred_depth, pred_depth_var, pred_logits = self.renderer(samples)
            loss, _terms = ops.tracking_losses(pred_color, pred_depth, pred_depth_var, pred_logits, samples["gt_color"],
                                               samples["gt_depth"], samples["gt_label"], samples["mask"],
                                               (self.lambda_p, self.lambda_d, self.lambda_l))
            with torch.no_grad():                       # keep-best without the host sync of :331
                better = loss < best_loss
                best_loss = torch.where(better, loss.detach(), best_loss)
                best_cam = torch.where(better, torch.cat((quad, T), 0).detach(), best_cam)
            loss.backward()
            optimizer.step()
        self.last_optimizer = optimizer
        return best_cam, best_loss

    def _track_frame_graphed(self, cur_frames, est_c2w, n_iters, features, warmup=0, refer_frames=None):
        from ._lib import ensure_init
        ensure_init()                                    # kernel attributes are set outside the capture (dns_init)
        self.static_shapes = True
        optimizer, quad, T = self.set_optimizer(est_c2w, fused=True)
        frames = dict(cur_frames)
        frames["est_quad"], frames["est_T"] = quad, T
        prep = self.prepare_frame(cur_frames)
        state = {"best_loss": torch.full((), float("inf"), device=self.device),
                 "best_cam": torch.cat((quad, T), 0).detach().clone()}

        def one_iter():
            optimizer.zero_grad(set_to_none=True)
            samples = self.get_target_samples(frames, refer_frames=refer_frames, features=features, prep=prep)
            pc, pd, pv, pl = self.renderer(samples)
            loss, _ = ops.tracking_losses(pc, pd, pv, pl, samples["gt_color"], samples["gt_depth"], samples["gt_label"],
                                          samples["mask"], (self.lambda_p, self.lambda_d, self.lambda_l))
            with torch.no_grad():
                better = loss < state["best_loss"]
                state["best_loss"].copy_(torch.where(better, loss.detach(), state["best_loss"]))
                state["best_cam"].copy_(torch.where(better, torch.cat((quad, T), 0).detach(), state["best_cam"]))
            loss.backward()
            optimizer.step()

        if warmup > 0:                                   # optional eager iterations on a side stream, then rewind
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            saved = (quad.detach().clone(), T.detach().clone())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    one_iter()
            torch.cuda.current_stream().wait_stream(side)
            with torch.no_grad():
                quad.copy_(saved[0]); T.copy_(saved[1])
                for m, v in optimizer.state.values():
                    m.zero_(); v.zero_()
                optimizer._dev_state.zero_()
                state["best_loss"].fill_(float("inf"))
                state["best_cam"].copy_(torch.cat((quad, T), 0))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):                        # records one iteration; nothing executes during capture
            one_iter()
        for _ in range(n_iters):                         # n_iters optimiser steps, like the eager loop and the reference
            g.replay()
        self.last_optimizer = optimizer
        return state["best_cam"], state["best_loss"]
