"""Mirror of the Mapper optimise-step bodies (reference slams/mapping.py) on the gfx950 kernels.

Same method names, argument meaning and return values as the reference for the part of ``Mapper`` that is
the hot path:

    set_decoder / fine_fn        slams/mapping.py:727-761, :590-601
    get_target_samples           :471-588   (2-D feature branch = an input here)
    renderer                     :603-635
    smoothness                   :129-159
    compute_*_loss               :110-126
    set_optimizer / optimize     :438-468, :839-949 (iteration driver)

Keyframe selection, visualisation, meshing, checkpointing and the process loop (``run``) are out of scope
(components 8-14).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from . import tcnn_shim as tcnn
from .common import feature_matching, get_opacity_loss, get_quad_from_c2w, get_rotation_from_quad
from .decoder import fused_cat


class FineDecoderPool(nn.Module):
    """The Mapper's ``fine_decoders`` dict (slams/mapping.py:736-749): one 80->33 network per semantic class,
    added lazily.  All weight sets live in ONE [capacity, count] buffer so the grouped MFMA kernel can switch
    sets per 128-point tile; ``self[class_id]`` is a ``tcnn_shim.Network`` whose ``.params`` is a view of its row,
    so optimisers, ``state_dict`` and pickling see ordinary per-class modules."""

    def __init__(self, n_in, n_out, net_cfg, capacity=64, device="cuda"):
        super().__init__()
        self.n_in, self.n_out, self.net_cfg = n_in, n_out, dict(net_cfg)
        self.nn_, self.nl = int(net_cfg["n_neurons"]), int(net_cfg["n_hidden_layers"])
        self.fp16 = str(net_cfg.get("dtype", "fp32")).lower() in ("fp16", "half", "float16")
        self.count = ops.mlp_param_count(n_in, n_out, self.nn_, self.nl)
        self.pool = nn.Parameter(torch.zeros(capacity, self.count, device=device))
        self.slot: Dict[int, int] = {}
        self._lut: Optional[torch.Tensor] = None

    def keys(self):
        return self.slot.keys()

    def __getitem__(self, class_id):
        """``fine_decoders[class_id]`` as the reference uses it (slams/mapping.py:600,1139): a Network whose flat
        ``params`` shares the pool row's storage (evaluation / inspection; training goes through the pool)."""
        net = tcnn.Network(self.n_in, self.n_out, self.net_cfg)
        net.params = nn.Parameter(self.pool.data[self.slot[int(class_id)]], requires_grad=False)
        return net

    def items(self):
        return [(c, self[c]) for c in self.slot]

    def __contains__(self, k):
        return int(k) in self.slot

    def __len__(self):
        return len(self.slot)

    def add(self, class_id: int, seed: int = 1337):
        class_id = int(class_id)
        if class_id in self.slot:
            return
        s = len(self.slot)
        if s >= self.pool.shape[0]:
            raise ValueError("FineDecoderPool capacity exceeded")
        init = tcnn.Network(self.n_in, self.n_out, self.net_cfg, seed=seed).params.data
        with torch.no_grad():
            self.pool[s].copy_(init.to(self.pool.device))
        self.slot[class_id] = s
        self._lut = None
        self.lut(0)                                       # host -> device copy NOW: never lazily inside a captured iteration

    def lut(self, max_class: int) -> torch.Tensor:
        """class id -> pool row (-1 = no decoder), on device."""
        if self._lut is None or self._lut.numel() <= max_class:
            n = max(max_class + 1, (max(self.slot) + 1) if self.slot else 1)
            l = torch.full((n,), -1, dtype=torch.int64)
            for c, s in self.slot.items():
                l[c] = s
            self._lut = l.to(self.pool.device)
        return self._lut

    def params_of(self, class_id: int) -> torch.Tensor:
        return self.pool[self.slot[int(class_id)]]

    def parameters_for(self, class_ids):
        # one tensor holds every class: the optimiser steps the pool (rows of absent classes get zero grad)
        return [self.pool]


class Mapper:
    """Hot-path subset of reference ``Mapper`` (slams/mapping.py:24-1146)."""

    def __init__(self, cfg: dict, decoder, bound: torch.Tensor, cam: dict, device="cuda",
                 label_layout: str = "reference_tiled"):
        self.cfg = cfg
        self.device = device
        self.decoder = decoder
        self.bound = bound.to(torch.float64)           # float64, as dns_slam.py:102-107
        self.bound_dev = self.bound.to(device)
        self.H, self.W = cam["H"], cam["W"]
        self.fx, self.fy, self.cx, self.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
        tr, mp = cfg["training"], cfg["mapping"]
        self.lr = tr["lr"]
        self.lambda_p, self.lambda_d, self.lambda_l = tr["lambda_color"], tr["lambda_depth"], tr["lambda_label"]
        self.lambda_sm, self.lambda_fs, self.lambda_opacity = tr["lambda_smooth"], tr["lambda_fs"], tr["lambda_opacity"]
        self.n_samples_ray, self.n_surface_ray = tr["n_samples_ray"], tr["n_surface_ray"]
        self.n_pixels = mp["n_pixels"]
        self.n_target_frame = mp["n_joint_optimize_frames"]
        self.BA_cam_lr = mp["BA_cam_lr"]
        self.start_optimize_idx = mp.get("start_optimize_idx", 10)
        self.hidden_dim = decoder.hidden_dim
        self.pe_dim, self.grid_dim = decoder.pe_dim, decoder.grid_dim
        self.label_layout = label_layout
        self.is_BA = True
        self.keyframe_dict, self.keyframe_list = [], []  # filled by the caller's frame loop (reference Mapper.run :975,1085)
        self.keyframe_selector = None                    # optional host-side keyframe choice (set_target_refer_frames)
        self.mapping_mode = "global"                     # 'global' | 'overlap' (slams/mapping.py:350-356; run() alternates them)
        self.encoder = None                              # frozen image stem (dns_slam_amd.encoder.ResNet) or None
        self.dist = None                                 # dns_slam_amd.dist.DistCtx for ray-batch data parallelism
        self.static_shapes = False                       # True: sync-free iteration (hipGraph-capturable)
        self.fused_losses = True                         # one HIP loss pass instead of ~60 torch launches
        self.use_map_step = False                        # True (+ static_shapes): optimize_frames runs fused_step.MapStep
        net_cfg = decoder.coarse_fn.decoder.network_config
        self.fine_decoders = FineDecoderPool(self.pe_dim + self.grid_dim, self.hidden_dim + 1, net_cfg, device=device)
        self.exist_decoders: Dict[int, int] = {}
        self.class2label_dict = cfg.get("class2label_dict", None)
        self.t_uniform = torch.linspace(0.0, 1.0, steps=self.n_samples_ray, device=device) if self.n_samples_ray > 0 else None
        # Device constants of the sync-free iteration are built HERE, not lazily in the first iteration: a pageable
        # host -> device copy (tensor.to(device)) is not capturable -- issued by the first iteration inside a hipGraph
        # capture it invalidates the capture (round-1 segfault in capture_end, DESIGN.md section 4 "stream capture").
        if str(device) != "cpu":
            from ._lib import ensure_init
            ensure_init()        # kernel attributes + the pinned error word NOW: the C side's lazy first-use init is refused
            #                      inside a stream capture, and PyTorch captures in global mode (any stream of the device)
            self._jitter_consts()
            self._ensure_lattice(tr.get("smooth_pts", 64), 0.1, 0.05)

    def _jitter_consts(self):
        ns = self.n_surface_ray
        if getattr(self, "_force_mask", None) is None or self._force_mask.numel() != ns:
            self._force_mask = torch.arange(ns, device=self.device) == (ns // 2 + 1)
            self._half = torch.full((ns,), 0.5, device=self.device)

    def _ensure_lattice(self, sample_points, voxel_size, margin):
        n = sample_points - 1
        key = (n, voxel_size, margin, str(self.device))
        if getattr(self, "_lattice_key", None) != key:
            bd = self.bound_dev
            grid_size = (sample_points - 1) * voxel_size
            self._offset_max = bd[:, 1] - bd[:, 0] - grid_size - 2 * margin                              # float64, on device
            ar = torch.arange(0, n, dtype=torch.long, device=self.device)
            gx, gy, gz = torch.meshgrid(ar, ar, ar, indexing="ij")
            self._lattice = torch.stack([gx, gy, gz], dim=-1).to(torch.float64)
            # the same elements in MORTON order of (i, j, k) and the inverse map: the sync-free smoothness path encodes the
            # lattice in that order (neighbouring rows share hash-table lines: the encoding kernel is 2.3x faster)
            def spread(v):
                v = v & 0x3ff
                v = (v | (v << 16)) & 0x30000ff
                v = (v | (v << 8)) & 0x300f00f
                v = (v | (v << 4)) & 0x30c30c3
                return (v | (v << 2)) & 0x9249249
            code = spread(gx.reshape(-1)) | (spread(gy.reshape(-1)) << 1) | (spread(gz.reshape(-1)) << 2)
            order = torch.argsort(code)
            self._lattice_morton = self._lattice.reshape(-1, 3)[order].contiguous()
            self._lattice_slot = torch.empty_like(order)
            self._lattice_slot[order] = torch.arange(order.numel(), device=self.device)
            d = bd[:, 1] - bd[:, 0]
            self._lattice_consts = (key, voxel_size / d, self._offset_max / d, margin / d)
            self._lattice_key = key
        return key

    # ------------------------------------------------------------------ losses (slams/mapping.py:110-126)
    def compute_photometric_loss(self, gt_color, pred_color):
        return ((gt_color - pred_color) ** 2).mean()

    def compute_depth_loss(self, gt_depth, pred_depth):
        mask = gt_depth > 0
        w = mask.float()
        cnt = w.sum()
        return (torch.abs(gt_depth - pred_depth) * w).sum() / cnt          # == mean over d>0, no host sync

    def compute_label_loss(self, gt_label, pred_logits):
        return F.cross_entropy(pred_logits, gt_label)

    def compute_latent_loss(self, coarse, fine):
        return ((coarse - fine) ** 2).mean()

    # ------------------------------------------------------------------ slams/mapping.py:129-159
    def smoothness(self, sample_points=64, voxel_size=0.1, margin=0.05, u_offset=None, u_jitter=None):
        n = sample_points - 1
        self._ensure_lattice(sample_points, voxel_size, margin)
        lattice, nx, halo = self._lattice, n, False
        if self.dist is not None and self.dist.union:
            # union-batch mode: this rank's x-planes [a, b) of the ONE lattice all ranks draw, plus the next slab's first plane
            # to close the x-differences (csrc/misc.hip); the ranks' partial sums add up to the cube's value
            if not self.fused_losses:
                raise ValueError("the lattice is only sharded on the fused-loss path")
            a, b = self.dist.shard(n)
            hi = min(b + 1, n)
            lattice, nx, halo = self._lattice[a:hi], hi - a, hi > b
        if u_offset is None and u_jitter is None and self.static_shapes:
            # sync-free iteration: the same fp64 affine map, folded to  pts = lattice * A + B  with
            # A = voxel / (b1 - b0),  B = (jitter * voxel + offset) / (b1 - b0)  -- 6 launches instead of 14
            _, c_vox, c_off, c_mar = self._lattice_consts
            r = getattr(self, "_lattice_r6", None) if getattr(self, "prefetch_draws", False) else None
            self._lattice_r6 = None
            if r is None:
                r = torch.rand(6, device=self.device)
            r = r.to(torch.float64)                                            # [offset(3) | jitter(3)], as float32 draws
            b = torch.addcmul(torch.addcmul(c_mar, r[:3], c_off), r[3:], c_vox)
            morton = self.fused_losses and lattice is self._lattice          # the whole cube (no rank slab): Morton-ordered rows
            pts = torch.addcmul(b, self._lattice_morton if morton else lattice, c_vox)
            pe, grid_pts = self.decoder.pe_fn(pts.reshape(-1, 3))
            if self.fused_losses:
                # Only the occupancy logit (output row 0, mapping.py:152) enters the TV term: the coarse network runs as an
                # (80 -> ... -> 1) network on the SAME parameter tensor (row 0 of W_out is the first row stored), so the other
                # 32 output rows are neither computed nor written, and no [n^3, 33] gradient full of zeros is built.
                net = self.decoder.coarse_fn.decoder
                occ = ops.mlp(fused_cat(pe, grid_pts), net.params, net.n_input_dims, 1, net.n_neurons, net.n_hidden_layers,
                              fp16=getattr(net, "fp16", False))
                if morton:
                    occ = occ.index_select(0, self._lattice_slot)           # back to the x-major cube for the TV kernel
                return ops.tv_smoothness(occ, n, sample_points, nx=nx, halo=halo)
            coarse = self.decoder.coarse_fn(pe, features=grid_pts)
            occ = coarse[:, 0:1].reshape(n, n, n, 1)
            tv_x = torch.pow(occ[1:, ...] - occ[:-1, ...], 2).sum()
            tv_y = torch.pow(occ[:, 1:, ...] - occ[:, :-1, ...], 2).sum()
            tv_z = torch.pow(occ[:, :, 1:, ...] - occ[:, :, :-1, ...], 2).sum()
            return (tv_x + tv_y + tv_z) / (sample_points ** 3)
        if u_offset is None:
            u_offset = torch.rand(3, device=self.device) if self.static_shapes else torch.rand(3)
        if u_jitter is None:
            u_jitter = torch.rand((1, 1, 1, 3), device=self.device) if self.static_shapes else torch.rand((1, 1, 1, 3))
        # reference: float64 through `volume`; (coords + jitter) * voxel + b0 + offset, then normalise by the bound
        bd = self.bound_dev
        offset = u_offset.to(self.device).to(torch.float64) * self._offset_max + margin
        pts = (lattice + u_jitter.to(self.device).to(torch.float64)) * voxel_size + bd[:, 0] + offset
        pts = (pts - bd[:, 0]) / (bd[:, 1] - bd[:, 0])
        shp = pts.shape
        pe, grid_pts = self.decoder.pe_fn(pts.reshape(-1, 3))
        coarse = self.decoder.coarse_fn(pe, features=grid_pts)
        if self.fused_losses:
            return ops.tv_smoothness(coarse, n, sample_points, nx=nx, halo=halo)        # :151-157 in one reduction kernel
        occ = coarse[:, 0:1].reshape(*shp[:3], 1)          # intended shape of :152
        tv_x = torch.pow(occ[1:, ...] - occ[:-1, ...], 2).sum()
        tv_y = torch.pow(occ[:, 1:, ...] - occ[:, :-1, ...], 2).sum()
        tv_z = torch.pow(occ[:, :, 1:, ...] - occ[:, :, :-1, ...], 2).sum()
        return (tv_x + tv_y + tv_z) / (sample_points ** 3)

    # ------------------------------------------------------------------ slams/mapping.py:727-761
    def set_decoder(self, target_frames):
        label_dict = target_frames["label_dict"]
        new_list = []
        for obj_id in label_dict:
            obj_id = int(obj_id)
            if self.class2label_dict is not None and obj_id not in self.class2label_dict:
                raise ValueError("Unknown semantic class", obj_id, "Existed class:", self.class2label_dict.keys())
            if obj_id not in self.fine_decoders:
                self.fine_decoders.add(obj_id)
                self.exist_decoders[obj_id] = 1
            else:
                self.exist_decoders[obj_id] += 1
            if self.exist_decoders[obj_id] <= 4:
                new_list.append(obj_id)
        min_obj = min(self.exist_decoders, key=self.exist_decoders.get)
        if min_obj not in new_list and self.exist_decoders[min_obj] < 10:
            self.exist_decoders[min_obj] += 1
            new_list.append(min_obj)
        return new_list

    # ------------------------------------------------------------------ slams/mapping.py:438-468
    def set_optimizer(self, target_frames, capturable=False, fused=False):
        net_para_list = list(self.decoder.parameters())
        net_para_list += self.fine_decoders.parameters_for(target_frames["label_dict"])
        quad_list, T_list = [], []
        for frame in range(self.n_target_frame):
            c2w = target_frames["est_c2w"][frame]
            quad = get_quad_from_c2w(c2w).clone().detach().to(self.device)
            T = c2w[:3, 3].clone().detach().to(self.device)
            if (self.n_target_frame == 1 or frame != 0) and self.is_BA:
                quad.requires_grad_(True)
                T.requires_grad_(True)
            quad_list.append(quad)
            T_list.append(T)
        net_para_list = [p for p in net_para_list if p.numel() > 0]
        if fused:
            from .optim import FusedAdam
            optimizer = FusedAdam([{"params": net_para_list, "lr": 0}, {"params": quad_list, "lr": 0},
                                   {"params": T_list, "lr": 0}])
            return optimizer, quad_list, T_list
        optimizer = torch.optim.Adam([{"params": net_para_list, "lr": 0},
                                      {"params": quad_list, "lr": 0},
                                      {"params": T_list, "lr": 0}], capturable=capturable)
        return optimizer, quad_list, T_list

    # ------------------------------------------------------------------ pixel picking (a1, a2)
    def prepare_frames(self, target_frames):
        """Once per optimize(): stack the K frames and build the per-frame class tables that
        ``select_by_class`` (utils/common.py:307-338) rebuilds with nonzero() every iteration."""
        K = self.n_target_frame
        color = torch.stack([target_frames["gt_color"][i] for i in range(K)]).to(self.device).float().contiguous()
        depth = torch.stack([target_frames["gt_depth"][i] for i in range(K)]).to(self.device).float().contiguous()
        label = torch.stack([target_frames["gt_label"][i] for i in range(K)]).to(self.device).float().contiguous()
        n_pixels = self.n_pixels // K
        n1, n2 = n_pixels // 3 * 2, n_pixels // 3           # slams/mapping.py:498,505
        if getattr(self, "rays_per_frame", None) is not None:  # benchmark override: exact (uniform, by-class) counts
            n1, n2 = self.rays_per_frame
        HW = self.H * self.W
        sorted_pix, starts, counts = [], [], []
        for f in range(K):
            lab = label[f].reshape(-1)
            order = torch.argsort(lab, stable=True)
            classes, cnt = torch.unique_consecutive(lab[order], return_counts=True)
            n_class = int(classes.numel())                       # one host sync per frame per optimize()
            n_k = n2 // n_class
            m = torch.full((n_class,), n_k, dtype=torch.int64)
            m[0] = n2 - n_k * (n_class - 1)
            st = torch.cumsum(cnt, 0) - cnt
            slot_class = torch.repeat_interleave(torch.arange(n_class), m).to(self.device)   # class of each of the n2 draws
            sorted_pix.append(order)
            starts.append(st[slot_class])
            counts.append(cnt[slot_class])
        # stacked tables: one draw for all K frames (8 launches per iteration instead of 8 per frame)
        offs = (torch.arange(K, device=self.device) * HW)[:, None]
        cnt_all = torch.stack(counts)
        return {"color": color, "depth": depth, "label": label, "n1": n1, "n2": n2, "HW": HW,
                "sorted_pix": sorted_pix, "starts": starts, "counts": counts,
                "sorted_flat": torch.stack(sorted_pix).reshape(-1), "starts_flat": torch.stack(starts) + offs,
                "counts_f64": cnt_all.to(torch.float64), "counts_m1": cnt_all - 1}

    # ------------------------------------------------------------------ draws one iteration ahead
    def _draw_all(self, prep):
        """The random draws of one static-shape iteration in the reference's order: pixels, surface jitter, lattice
        offset / jitter (mapping.py:498-505, common.py:571-582, mapping.py:137-143)."""
        return {"pix": self.draw_pixels(prep), "jitter": self.draw_jitter(), "r6": torch.rand(6, device=self.device)}

    def _take_draws(self, prep):
        """``prefetch_draws``: the dozen tiny generator / index launches of iteration k+1 are enqueued on the side
        stream while iteration k runs, instead of heading k+1's critical path (same generator, same order, same
        values -- the host issues them in sequence either way).  Needs ``static_shapes`` and a smoothness term in every
        iteration (the lattice draw is taken with the others)."""
        main = torch.cuda.current_stream()
        pend = getattr(self, "_pending_draws", None)
        if pend is not None and pend[2] is not prep:
            pend = None                                  # drawn from another optimize()'s class tables: drop them
        if pend is None:
            cur = self._draw_all(prep)
        else:
            cur, ev, _ = pend
            main.wait_event(ev)
        if getattr(self, "_side_stream", None) is None:
            self._side_stream = torch.cuda.Stream(device=self.device)
        side = self._side_stream
        # Every burst of side-stream work starts by waiting for the main stream's tail: tensors the side stream allocated
        # earlier and the main stream consumed (the smoothness loss, gradients) carry no record_stream, so their memory
        # may only be handed out again once the main stream is past its reads.  (Without this wait the draws landed in
        # such blocks while the main stream was still a step behind: measured as run-to-run different losses.)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            nxt = self._draw_all(prep)
            ev = torch.cuda.Event()
            ev.record(side)
        for t in (nxt["pix"], nxt["jitter"][0], nxt["jitter"][1], nxt["r6"]):
            t.record_stream(main)
        self._pending_draws = (nxt, ev, prep)
        self._lattice_r6 = cur["r6"]
        return cur

    def draw_pixels(self, prep):
        """Indices of one iteration: per frame n1 uniform picks (select_uv, common.py:274) then n2 class-balanced
        picks (select_by_class :313-328), concatenated in the reference's order."""
        K = self.n_target_frame
        i1 = torch.randint(prep["HW"], (K, prep["n1"]), device=self.device)
        u = torch.rand(K, prep["n2"], device=self.device, dtype=torch.float64)
        j = torch.minimum((u * prep["counts_f64"]).to(torch.int64), prep["counts_m1"])
        i2 = prep["sorted_flat"][prep["starts_flat"] + j]
        return torch.cat((i1, i2), 1).reshape(-1)

    def draw_jitter(self, n_frames=None):
        """The two draws of sample_along_rays (utils/common.py:571-574,582), forced 0.5 included, ONE PAIR PER FRAME as in
        the reference (get_target_samples calls sample_along_rays per target frame, slams/mapping.py:530): returns
        (t_surf, t_zero), each [n_frames, n_surface]; the K frames still sample in one launch (the kernel reads its frame's
        rows).  ``self.shared_jitter = True`` returns one [n_surface] pair shared by the frames (round-1 behaviour, kept as
        a benchmark option).  Reference: CPU generator then .to(device); with ``static_shapes`` the draws come from the
        device generator (no host work: the iteration can be captured in a hipGraph) and the forced 0.5 is a select."""
        ns = self.n_surface_ray
        K = self.n_target_frame if n_frames is None else n_frames
        shared = getattr(self, "shared_jitter", False)
        if self.static_shapes:
            self._jitter_consts()
            # the reference forces t[ns//2+1] = 0.5 only when no draw equals 0.5 exactly (probability ~2e-6 per call with
            # float32 draws); the sync-free path always forces it: one select instead of a reduction + three more launches
            if shared:
                r = torch.rand(2, ns, device=self.device)
                return torch.where(self._force_mask, self._half, r[0]), r[1]
            r = torch.rand(2, K, ns, device=self.device)
            return torch.where(self._force_mask, self._half, r[0]), r[1]
        ts, t0s = [], []
        for _ in range(1 if shared else K):                 # per frame: rand(ns), forced 0.5, rand(ns) -- the reference's order
            t = torch.rand(ns)
            if not torch.any(t == 0.5):
                t[ns // 2 + 1] = 0.5
            ts.append(t)
            t0s.append(torch.rand(ns))
        if shared:
            return ts[0].to(self.device), t0s[0].to(self.device)
        return torch.stack(ts).to(self.device), torch.stack(t0s).to(self.device)

    def _check_same_draws(self, full):
        """Union-batch mode relies on identical generator states on all ranks: verified once (one tiny all-reduce)."""
        if getattr(self, "_union_checked", False):
            return
        s = full.sum().to(torch.float64)
        pair = torch.stack((s, -s))
        self.dist.allreduce_max(pair)
        if float(pair[0]) != -float(pair[1]):
            raise RuntimeError("union-batch mode: the ranks drew different pixel lists -- seed every rank identically "
                               "(torch.manual_seed + torch.cuda.manual_seed) before the first iteration")
        self._union_checked = True

    # ------------------------------------------------------------------ slams/mapping.py:471-588
    def get_target_samples(self, target_frames, quad_list, T_list, refer_frames=None, features=None,
                           prep=None, pix_idx=None, jitter=None):
        """All K frames in one launch.  ``jitter`` = (t_surf, t_zero), each [K, n_surface] (one pair per frame, as the
        reference draws them) or [n_surface] (shared by the frames); a list of K pairs is accepted too."""
        if prep is None:
            prep = self.prepare_frames(target_frames)
        K = self.n_target_frame
        if pix_idx is None and jitter is None and getattr(self, "prefetch_draws", False) and self.static_shapes:
            d = self._take_draws(prep)
            pix_idx, jitter = d["pix"], d["jitter"]
        if pix_idx is None:
            pix_idx = self.draw_pixels(prep)
        npf = pix_idx.numel() // K
        quat = torch.stack(list(quad_list))
        trans = torch.stack(list(T_list))
        cam = (self.fx, self.fy, self.cx, self.cy)
        window = (0, self.H, 0, self.W)
        if jitter is None:
            jitter = self.draw_jitter()
        if isinstance(jitter, (list,)) and len(jitter) == K and isinstance(jitter[0], (tuple, list)):
            jitter = (torch.stack([j[0] for j in jitter]), torch.stack([j[1] for j in jitter]))
        dmax = None
        point_labels = None
        if self.dist is not None and self.dist.union:
            # union-batch mode (dist.py): pix_idx / jitter are the SAME on every rank; this rank renders rays [a, b) of every
            # frame's list.  The batch-global max(gt_depth) (utils/common.py:581,591) comes from the whole list: no collective.
            full = pix_idx.reshape(K, npf)
            self._check_same_draws(full)
            a, b = self.dist.shard(npf)
            dmax = torch.gather(prep["depth"].reshape(K, -1), 1, full).amax(dim=1).clamp_min(0.0)
            if self.label_layout == "reference_tiled":
                # point k of the (union) batch is routed by label[k mod N] -- a function of the GLOBAL ray index
                if not self.static_shapes:
                    raise ValueError("union-batch mode with the reference_tiled label layout needs static_shapes (no ray drop)")
                from .dist import union_point_labels
                S = self.t_uniform.numel() + self.n_surface_ray
                lab = torch.gather(prep["label"].reshape(K, -1), 1, full).reshape(-1).long()
                point_labels = union_point_labels(lab, K, npf, a, b, S)
            if features is not None and features.dim() == 3:
                features = features.reshape(K, npf, *features.shape[1:])[:, a:b].reshape(K * (b - a), *features.shape[1:])
            pix_idx, npf = full[:, a:b].reshape(-1), b - a
        elif self.dist is not None and self.dist.enabled:
            # batch-global max(gt_depth) of sample_along_rays (utils/common.py:581,591): over ALL ranks' rays of a frame
            dflat = prep["depth"].reshape(K, -1)
            dmax = torch.gather(dflat, 1, pix_idx.reshape(K, npf)).amax(dim=1).clamp_min(0.0)
            self.dist.allreduce_max(dmax)
        res = ops.raygen_sample(quat, trans, pix_idx, prep["color"], prep["depth"], prep["label"], cam, self.bound,
                                window, npf, self.t_uniform, jitter[0].contiguous(), jitter[1].contiguous(), depth_max=dmax)
        rays_o, rays_d, pts, gt_color, gt_depth, gt_label, inside, z = res
        N, S = z.shape
        if features is None:
            # no 2-D branch: one read-only block of zeros, made once per shape instead of a 33 MB fill per iteration
            code = getattr(self, "_zero_code", None)
            if code is None or code.shape != (N, S, self.hidden_dim):
                code = self._zero_code = torch.zeros(N, S, self.hidden_dim, device=self.device)
        else:
            if features.dim() == 5:
                # stem feature maps [n_target, n_refer, C, h, w] + refer_frames: the 2-D branch of :533-551
                features = self.match_features(target_frames, quad_list, T_list, refer_frames, features, pts, npf)
            d = gt_depth[:, None]
            front = (z < d * 0.95).float()
            back = (z > d * 1.05).float()
            trunc = (1.0 - front) * (1.0 - back) * (d > 0.0).float()        # :553-556
            code = features * trunc[..., None]
        if self.static_shapes:
            # no compaction (and no host sync): rays leaving the box stay in the batch with valid = 0 and are
            # excluded from every loss mean -- same values as dropping them (slams/mapping.py:576-586) for the
            # per_ray label layout; the reference_tiled layout (D1) depends on the post-drop N and needs the sync path
            out = {"gt_color": gt_color, "gt_depth": gt_depth, "gt_label": gt_label, "rays_o": rays_o, "rays_d": rays_d,
                   "pts": pts, "z_vals": z, "features": code, "valid": inside}
            if point_labels is not None:
                out["point_labels"] = point_labels
            return out
        mask = inside.bool()
        if bool(mask.all()):                                                # the reference syncs here too (:576)
            sel = lambda t: t
        else:
            sel = lambda t: t[mask]
        return {"gt_color": sel(gt_color), "gt_depth": sel(gt_depth), "gt_label": sel(gt_label),
                "rays_o": sel(rays_o), "rays_d": sel(rays_d), "pts": sel(pts), "z_vals": sel(z), "features": sel(code)}

    # ------------------------------------------------------------------ slams/mapping.py:533-551
    def match_features(self, target_frames, quad_list, T_list, refer_frames, features, pts, npf):
        """Per target frame: choose each reference frame's pose (-1 = the frame itself, a frame that is also a target
        = its pose under optimisation, else the stored keyframe pose; all detached, :534-545), project the frame's points
        and merge the looked-up image codes (feature_matching + Decoder.merge).  -> code [N, S, hidden_dim]."""
        K = torch.tensor([[self.fx, 0.0, self.cx], [0.0, self.fy, self.cy], [0.0, 0.0, 1.0]])
        bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=self.device)
        target_idx = list(target_frames["kf_idx"])
        S = pts.shape[1]
        out = []
        for i in range(self.n_target_frame):
            def pose_of(k):
                R = get_rotation_from_quad(quad_list[k])
                return torch.cat([torch.cat((R, T_list[k][:, None]), -1), bottom], dim=0).clone().detach()
            refer_idx = list(refer_frames["kf_idx"][i])
            w2c = []
            for refer_id in refer_idx:
                if refer_id == -1:
                    c2w = pose_of(i)
                elif refer_id in target_idx:
                    c2w = pose_of(target_idx.index(refer_id))
                else:
                    c2w = refer_frames["est_c2w"][i][refer_idx.index(refer_id)].clone().detach().to(self.device).float()
                w2c.append(torch.inverse(c2w))
            w2c = torch.stack(w2c, 0)
            p_i = pts[i * npf:(i + 1) * npf].flatten(0, 1)
            code = feature_matching(self.H, self.W, K, p_i, w2c, features[i], self.decoder.merge)
            out.append(code.reshape(npf, S, -1))
        return torch.cat(out, 0)

    # ------------------------------------------------------------------ slams/mapping.py:590-601
    def _class_slots(self, classes, strict):
        """class id -> row of the pooled per-class parameters (-1: no network); ``strict`` mirrors the reference's
        ValueError on an unknown class (one host sync)."""
        lut = self.fine_decoders.lut(0)
        # lut padded with -1 on both sides: one clamp + one gather instead of two compares, an and, a fill and a where
        if getattr(self, "_lut_ext_src", None) is not lut:
            pad = torch.full((1,), -1, dtype=lut.dtype, device=lut.device)
            self._lut_ext, self._lut_ext_src = torch.cat((pad, lut, pad)), lut
        slot = self._lut_ext[classes.clamp(min=-1, max=lut.numel()) + 1]
        if strict and bool((slot < 0).any()):
            missing = torch.unique(classes[slot < 0]).tolist()
            raise ValueError("Fine decoders does NOT have class", missing)
        return slot

    def fine_fn(self, pes, classes=None, features=None, strict=True):
        slot = self._class_slots(classes, strict)
        x = fused_cat(pes, features)
        n = len(self.fine_decoders)
        return ops.mlp_grouped(x, self.fine_decoders.pool[:max(n, 1)], slot, self.pe_dim + self.grid_dim,
                               self.hidden_dim + 1, self.fine_decoders.nn_, self.fine_decoders.nl,
                               fp16=getattr(self.fine_decoders, "fp16", False))

    # ------------------------------------------------------------------ slams/mapping.py:603-635
    def renderer(self, samples, strict=True, need_coarse=True):
        """``need_coarse=False`` (forward-only callers that drop the coarse latents, e.g. ``render_frame``): the coarse network is
        not run and the sixth return value is None."""
        pts = samples["pts"]
        n_pts, n_samples, _ = pts.shape
        z_vals = samples["z_vals"]
        gt_label = samples["gt_label"]
        slot_rays = None
        if samples.get("point_labels") is not None:
            classes = samples["point_labels"]                              # union-batch shard: the tiling of the WHOLE batch
        elif self.label_layout == "reference_tiled":
            classes = None if gt_label.dim() == 1 else gt_label.repeat(1, n_samples).flatten(0, 1)   # :613 -- tiles, SURVEY D1
            slot_rays = gt_label if gt_label.dim() == 1 else None
        else:
            classes = gt_label.repeat_interleave(n_samples)
        pixel_pts = samples["features"].flatten(0, 1)
        buf = self.decoder.pe_fn.forward_world(pts.flatten(0, 1), self.bound)     # :608 + pe_fn, fused
        pe, grid_pts = buf[:, :self.pe_dim], buf[:, self.pe_dim:]
        if getattr(self, "fused_nets", True) and self.pe_dim % 4 == 0 and self.pe_dim <= 64 and \
                self.hidden_dim + pixel_pts.shape[1] <= 64 and (self.hidden_dim + pixel_pts.shape[1]) % 4 == 0:
            # the four networks as one autograd node: no cat for the colour / logit input, in-place gradient sums
            if slot_rays is not None:
                # the class -> decoder-row lookup per RAY, then tiled like the labels (the same values as looking up the tiled
                # labels: 1 / n_samples of the clamp / add / gather work -- a frame render's chunks hold 4.2 M points)
                slot = self._class_slots(slot_rays, strict).repeat(n_samples)
            else:
                slot = self._class_slots(classes, strict)
            dec, pool = self.decoder, self.fine_decoders
            net = lambda m: (m.n_input_dims, m.n_output_dims, m.n_neurons, m.n_hidden_layers)
            coarse_latents, fine_latents, values_pts, logits_pts = ops.render_nets(
                buf, pixel_pts, dec.coarse_fn.decoder.params, pool.pool,
                dec.out_fn.color_decoder.params, dec.out_fn.logit_decoder.params, slot, self.pe_dim,
                net(dec.coarse_fn.decoder), (self.pe_dim + self.grid_dim, self.hidden_dim + 1, pool.nn_, pool.nl),
                net(dec.out_fn.color_decoder), net(dec.out_fn.logit_decoder),
                fp16=getattr(dec.coarse_fn.decoder, "fp16", False), n_groups=max(len(pool), 1),
                need_coarse=need_coarse or torch.is_grad_enabled())
            if not (need_coarse or torch.is_grad_enabled()):
                coarse_latents = None
            # values_pts = (sigmoid colour | occupancy)
        else:
            if classes is None:
                classes = gt_label.repeat(1, n_samples).flatten(0, 1)
            coarse_latents = self.decoder.coarse_fn(pe, features=grid_pts) if (need_coarse or torch.is_grad_enabled()) else None
            fine_latents = self.fine_fn(pe, classes=classes, features=grid_pts, strict=strict)
            color_pts, logits_pts = self.decoder.out_fn(pe, torch.cat((fine_latents[:, 1:], pixel_pts), -1))
            values_pts = torch.cat((color_pts, fine_latents[:, 0:1]), -1)
        values_pts = values_pts.reshape(n_pts, n_samples, -1)
        logits_pts = logits_pts.reshape(n_pts, n_samples, -1)
        pred_depth, pred_depth_var, pred_color, weights, pred_logits = ops.composite(values_pts, z_vals, logits_pts)
        return pred_color, pred_depth, pred_depth_var, pred_logits, fine_latents, coarse_latents

    # ------------------------------------------------------------------ slams/meshing.py:461-503
    @torch.no_grad()
    def eval_points(self, pts, pixel_pts=None, gt_label_pts=None, stage="fine", n_pts_batch=1 << 20):
        """``Mesher.eval_points`` (the meshing / evaluation query; decoders = this mapper's): world points [P,3] ->
        (values [P,4] = sigmoid colour + occupancy logit, -100 outside the open bound; labels [P] = argmax of the logit
        network, -1 outside; ``None`` for ``stage='coarse'``).  ``gt_label_pts`` routes the fine decoders exactly like
        the reference (an unknown class raises ValueError, classes with a single point give zeros).  Forward only; runs
        in chunks of ``n_pts_batch`` points (the reference's marching-cubes driver sends 500 000 at a time)."""
        dev = self.device
        pts = pts.to(dev).float()
        P = pts.shape[0]
        pixel = None if pixel_pts is None else pixel_pts.to(dev).float()          # (None: zeros, built only where a kernel needs them)
        if stage != "coarse" and gt_label_pts is None:
            raise ValueError("eval_points(stage='fine') needs gt_label_pts")
        classes = None if stage == "coarse" else gt_label_pts.to(dev).long().reshape(-1)
        b = self.bound_dev
        p64 = pts.to(torch.float64)
        mask = ((p64[:, 0] < b[0, 1]) & (p64[:, 0] > b[0, 0]) & (p64[:, 1] < b[1, 1]) & (p64[:, 1] > b[1, 0]) &
                (p64[:, 2] < b[2, 1]) & (p64[:, 2] > b[2, 0]))
        if classes is not None:
            slot_all = self._class_slots(classes, True)       # unknown class -> ValueError, like meshing.py:451
            # the > 1 point rule is per CALL in the reference: count over all points, not per chunk
            cnt_all = torch.bincount(slot_all.clamp_min(0), minlength=max(len(self.fine_decoders), 1))
        values, labels = [], []
        n_code = 0 if pixel_pts is None else pixel.shape[1]
        fused = getattr(self, "fused_nets", True) and self.pe_dim % 4 == 0 and self.pe_dim <= 64 and \
            self.hidden_dim + n_code <= 64 and (self.hidden_dim + n_code) % 4 == 0 and (self.pe_dim + self.hidden_dim) % 8 == 0
        dec = self.decoder
        net = lambda m: (m.n_input_dims, m.n_output_dims, m.n_neurons, m.n_hidden_layers)
        for s0 in range(0, P, n_pts_batch):
            s1 = min(s0 + n_pts_batch, P)
            buf = self.decoder.pe_fn.forward_world(pts[s0:s1], self.bound)
            pe, grid_pts = buf[:, :self.pe_dim], buf[:, self.pe_dim:]
            if fused:
                # The renderer's fused node, forward only (round 5; this query ran 1.65 ns per point against the frame render's
                # 0.72): no cat of the encoder's two halves, no [P, 64] feature block, the colour network writes the [P, 4] rows;
                # without a 2-D code (pixel_pts None: a code of ZERO columns) the colour / logit networks run as their live
                # (OneBlob + latent)-input networks and read the latent straight out of the latent rows.  stage='coarse': the
                # coarse network takes the fine decoders' place as a pool of one weight set.
                if classes is None:
                    cdec = dec.coarse_fn.decoder
                    pool_p, shp_f, n_g = cdec.params.reshape(1, -1), net(cdec), 1
                    slot = torch.zeros(s1 - s0, dtype=torch.int64, device=dev)
                    fp16 = getattr(cdec, "fp16", False)
                else:
                    pool = self.fine_decoders
                    pool_p, n_g = pool.pool, max(len(pool), 1)
                    shp_f = (self.pe_dim + self.grid_dim, self.hidden_dim + 1, pool.nn_, pool.nl)
                    sl = slot_all[s0:s1]
                    slot = torch.where((sl >= 0) & (cnt_all[sl.clamp_min(0)] > 1), sl, torch.full_like(sl, -1))
                    fp16 = getattr(dec.coarse_fn.decoder, "fp16", False)
                code = torch.empty(s1 - s0, 0, device=dev) if pixel_pts is None else pixel[s0:s1]
                _, _, raw, logits = ops.render_nets(buf, code, dec.coarse_fn.decoder.params, pool_p, dec.out_fn.color_decoder.params,
                                                    dec.out_fn.logit_decoder.params, slot, self.pe_dim, net(dec.coarse_fn.decoder),
                                                    shp_f, net(dec.out_fn.color_decoder), net(dec.out_fn.logit_decoder), min_count=1,
                                                    fp16=fp16, n_groups=n_g, need_coarse=False)
                values.append(raw)
                if classes is not None:
                    labels.append(torch.argmax(logits, dim=-1))
                continue
            if pixel is None:
                pixel = torch.zeros(P, self.hidden_dim, device=dev)
            if classes is None:
                lat = self.decoder.coarse_fn(pe, features=grid_pts)
            else:
                pool = self.fine_decoders
                cnt = cnt_all
                slot = torch.where((slot_all[s0:s1] >= 0) & (cnt[slot_all[s0:s1].clamp_min(0)] > 1), slot_all[s0:s1],
                                   torch.full_like(slot_all[s0:s1], -1))
                lat = ops.mlp_grouped(fused_cat(pe, grid_pts), pool.pool[:max(len(pool), 1)], slot, self.pe_dim + self.grid_dim,
                                      self.hidden_dim + 1, pool.nn_, pool.nl, min_count=1, fp16=getattr(pool, "fp16", False))
            color, logits = self.decoder.out_fn(pe, torch.cat((lat[:, 1:], pixel[s0:s1]), -1))
            values.append(torch.cat((color, lat[:, 0:1]), -1))
            if classes is not None:
                labels.append(torch.argmax(logits, dim=-1))
        values = torch.cat(values) if len(values) != 1 else values[0]
        values[~mask, 3] = -100
        if classes is None:
            return values, None
        labels = torch.cat(labels) if len(labels) != 1 else labels[0]
        labels[~mask] = -1
        return values, labels

    # ------------------------------------------------------------------ slams/mapping.py:638-724 (without the plotting)
    @torch.no_grad()
    def render_frame(self, cur_gt_color, cur_gt_depth, cur_gt_label, cur_c2w, features=None, n_pts_batch=None, jitter=None):
        """Full-image re-rendering of ``frame_vis``: all H*W rays of the frame (get_all_rays :540, box clip, ONE
        sample_along_rays over the whole image :661), then the renderer in chunks of ``n_pts_batch`` rays (:675-694;
        the reference's 1000 by default -- NB the tiled label layout (D1) makes the result depend on the chunk size).
        Returns (pred_color [H,W,3], pred_depth [H,W], pred_label [H,W] = argmax of the composited logits)."""
        H, W = self.H, self.W
        n_pts_batch = n_pts_batch or self.cfg["mapping"].get("n_pts_batch", 1000)
        dev = self.device
        quat = get_quad_from_c2w(cur_c2w).to(dev)[None]
        trans = cur_c2w[:3, 3].to(dev).float()[None]
        f = lambda t: t.to(dev).float().contiguous()[None]
        if jitter is None:
            jitter = self.draw_jitter(1)
        pix = torch.arange(H * W, device=dev)
        rays_o, rays_d, pts, gt_color, gt_depth, gt_label, inside, z = ops.raygen_sample(
            quat, trans, pix, f(cur_gt_color), f(cur_gt_depth), f(cur_gt_label), (self.fx, self.fy, self.cx, self.cy),
            self.bound, (0, H, 0, W), H * W, self.t_uniform, jitter[0], jitter[1])
        S = z.shape[1]
        colors, depths, labels = [], [], []
        for start in range(0, H * W, n_pts_batch):
            end = min(start + n_pts_batch, H * W)
            if features is None:
                # no 2-D code: a code of ZERO columns -- the colour / logit networks run as their live (OneBlob + latent)-input
                # networks (the reference multiplies a zero code through, :553-557; round 4 read a 537 MB block of zeros per chunk)
                code = torch.empty(end - start, S, 0, device=dev)
            else:
                code = features[start:end]
            samples = {"rays_o": rays_o[start:end], "rays_d": rays_d[start:end], "gt_label": gt_label[start:end],
                       "pts": pts[start:end], "z_vals": z[start:end], "features": code}
            color, depth, _, logits, _, _ = self.renderer(samples, strict=False, need_coarse=False)
            colors.append(color), depths.append(depth), labels.append(torch.argmax(logits, dim=-1))
        return torch.cat(colors).reshape(H, W, 3), torch.cat(depths).reshape(H, W), torch.cat(labels).reshape(H, W)

    # ------------------------------------------------------------------ slams/mapping.py:887-907
    def iteration_loss(self, samples, lambda_lt=10.0, smooth=True, u_offset=None, u_jitter=None, strict=False):
        # The smoothness branch (lattice encode + coarse network, forward and -- since autograd replays a node on its
        # forward stream -- backward) shares nothing with the ray branch until the gradients meet: on a second HIP
        # stream its kernels fill the ramps / tails of the ray branch's launches (both capture into one hipGraph).
        side = None
        if smooth and getattr(self, "overlap_smooth", False) and self.device != "cpu":
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream(device=self.device)
            side = self._side_stream
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                smooth_pre = self.smoothness(sample_points=self.cfg["training"]["smooth_pts"], u_offset=u_offset,
                                             u_jitter=u_jitter)
        pred_color, pred_depth, _, pred_logits, fine_latents, coarse_latents = self.renderer(samples, strict=strict)
        tr = self.cfg["training"]
        world = self.dist.world_size if (self.dist is not None and self.dist.enabled) else 1
        if self.fused_losses:
            lam = (self.lambda_p, self.lambda_d, self.lambda_l, lambda_lt, self.lambda_fs, self.lambda_opacity,
                   tr["opacity_sigma"], 0.05)          # opacity_sigma lands in `truncation`, sigma stays 0.05 (D5)
            red = self.dist.allreduce_sums if world > 1 else None
            loss, t = ops.mapping_losses(pred_color, pred_depth, pred_logits, fine_latents, coarse_latents,
                                         samples["gt_color"], samples["gt_depth"], samples["gt_label"], samples["z_vals"],
                                         lam, valid=samples.get("valid"), reduce_sums=red)
            terms = {"p_loss": t[0], "d_loss": t[1], "l_loss": t[2], "lt_loss": t[3], "fs_loss": t[4], "opacity_loss": t[5]}
        else:
            if samples.get("valid") is not None or world > 1:
                raise ValueError("the unfused loss path supports neither static_shapes nor multi-GPU")
            d_loss = self.compute_depth_loss(samples["gt_depth"], pred_depth)
            p_loss = self.compute_photometric_loss(samples["gt_color"], pred_color)
            l_loss = self.compute_label_loss(samples["gt_label"], pred_logits)
            lt_loss = self.compute_latent_loss(coarse_latents, fine_latents)
            fs_loss, opacity_loss = get_opacity_loss(samples["z_vals"], samples["gt_depth"], fine_latents[..., -1],
                                                     tr["opacity_sigma"])
            loss = self.lambda_p * p_loss + self.lambda_d * d_loss + self.lambda_l * l_loss + lambda_lt * lt_loss + \
                self.lambda_fs * fs_loss + self.lambda_opacity * opacity_loss
            terms = {"p_loss": p_loss, "d_loss": d_loss, "l_loss": l_loss, "lt_loss": lt_loss, "fs_loss": fs_loss,
                     "opacity_loss": opacity_loss}
        if smooth:
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)
                smooth_loss = smooth_pre
            else:
                smooth_loss = self.smoothness(sample_points=tr["smooth_pts"], u_offset=u_offset, u_jitter=u_jitter)
            # multi-GPU: gradients are SUMMED over ranks.  weak mode: every rank evaluates its own lattice, each contributes 1/W
            # (average of W lattices); union mode: the ranks hold slabs of ONE lattice whose values add up to the cube's
            w_sm = self.lambda_sm if (world > 1 and self.dist.union) else self.lambda_sm / world
            loss = loss + w_sm * smooth_loss
            terms["smooth_loss"] = smooth_loss
        return loss, terms

    # ------------------------------------------------------------------ slams/mapping.py:764-836
    def decoder_init(self, decoder_idx, gt_color, gt_depth, gt_label, gt_c2w, cur_c2w, n_iters=100, n_rays=300,
                     features=None, smooth=True):
        """Warm-up of freshly added per-class decoders on the current frame: ``n_iters`` Adam steps (lr = training.lr) on
        (decoder + fine decoders) with ``n_rays`` rays drawn class-balanced over ``decoder_idx`` only
        (get_samples_by_uniq_class, utils/common.py:364-403: first class takes the remainder, a 1-pixel class is
        repeated, an absent class is skipped), fixed pose, losses = depth + colour + label + smoothness + fs/opacity
        (no latent term, :823-824)."""
        dev = self.device
        color = gt_color.to(dev).float().contiguous()[None]
        depth = gt_depth.to(dev).float().contiguous()[None]
        label = gt_label.to(dev).float().contiguous()[None]
        lab = label[0].reshape(-1)
        order = torch.argsort(lab, stable=True)
        classes, cnt = torch.unique_consecutive(lab[order], return_counts=True)
        st = torch.cumsum(cnt, 0) - cnt
        cls_host = classes.tolist()                                       # one sync per decoder_init call
        n_class = len(decoder_idx)
        n_k = n_rays // n_class
        starts, counts = [], []
        for i, c in enumerate(decoder_idx):
            m = n_rays - n_k * (n_class - 1) if i == 0 else n_k
            if float(c) in cls_host:
                j = cls_host.index(float(c))
                starts.append(st[j].expand(m))
                counts.append(cnt[j].expand(m))
        starts, counts = torch.cat(starts), torch.cat(counts)
        quat = get_quad_from_c2w(cur_c2w).to(dev)[None]
        trans = cur_c2w[:3, 3].to(dev).float()[None]
        from .optim import FusedAdam
        params = [p for p in list(self.decoder.parameters()) + [self.fine_decoders.pool] if p.numel() > 0]
        optimizer = FusedAdam([{"params": params, "lr": self.lr}])
        tr = self.cfg["training"]
        lam = (self.lambda_p, self.lambda_d, self.lambda_l, 0.0, self.lambda_fs, self.lambda_opacity, tr["opacity_sigma"], 0.05)
        loss = None
        for _ in range(n_iters):
            optimizer.zero_grad()
            u = torch.rand(starts.numel(), device=dev, dtype=torch.float64)
            pix = order[starts + torch.minimum((u * counts).to(torch.int64), counts - 1)]
            jit = self.draw_jitter(1)
            rays_o, rays_d, pts, gc, gd, gl, inside, z = ops.raygen_sample(
                quat, trans, pix, color, depth, label, (self.fx, self.fy, self.cx, self.cy), self.bound,
                (0, self.H, 0, self.W), pix.numel(), self.t_uniform, jit[0], jit[1])
            code = torch.zeros(z.shape[0], z.shape[1], self.hidden_dim, device=dev) if features is None else features
            samples = {"pts": pts, "z_vals": z, "gt_label": gl, "features": code, "rays_d": rays_d}
            pc, pd, _, pl, fine, coarse = self.renderer(samples, strict=True)
            loss, _t = ops.mapping_losses(pc, pd, pl, fine, coarse, gc, gd, gl, z, lam)
            if smooth:
                loss = loss + self.lambda_sm * self.smoothness(sample_points=tr["smooth_pts"])
            loss.backward()
            optimizer.step()
        return loss

    # ------------------------------------------------------------------ slams/mapping.py:171-236
    def keyframe_selection_overlap(self, gt_color, gt_depth, c2w, keyframe_dict, k, N_samples=16, pixels=100, th=0.0):
        """The reference's overlap test: ``pixels`` rays of the current frame (get_samples: one uniform draw on the device
        generator, :199), ``N_samples`` points per ray between 0.8 d and d + 0.5, projected into every keyframe's estimated
        camera (x flipped, K, :223-226); a keyframe's score is the fraction of the points that land more than 10 px inside
        its image with z < 0 (:228-232).  Keyframes scoring above ``th``, most overlap first, are shuffled with
        ``np.random.permutation`` and the first k are returned (:235-239) -- the reference's draw order, so a seeded run picks
        the same keyframes.  Host-side control flow around the hot path (two tiny device kernels + one [n_kf] read-back)."""
        import numpy as np
        from .common import get_samples
        dev = self.device
        H, W, fx, fy, cx, cy = self.H, self.W, self.fx, self.fy, self.cx, self.cy
        c2w = torch.as_tensor(c2w).to(dev).float()
        img = torch.cat((gt_color.to(dev).float(), gt_depth.to(dev).float().unsqueeze(-1)), -1)
        rays_o, rays_d, samples = get_samples(0, H, 0, W, pixels, H, W, fx, fy, cx, cy, c2w[:3, :3], c2w[:3, -1], img, dev)
        d = samples[:, -1].reshape(-1, 1)
        t_vals = torch.linspace(0.0, 1.0, steps=N_samples, device=dev)
        z_vals = (d * 0.8) * (1.0 - t_vals) + (d + 0.5) * t_vals
        pts = (rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., None]).reshape(-1, 3)              # [pixels * N_samples, 3]
        if len(keyframe_dict) == 0:
            return []
        w2c = torch.linalg.inv(torch.stack([torch.as_tensor(kf["est_c2w"]).to(dev).float() for kf in keyframe_dict]))
        cam = torch.einsum("kij,nj->kni", w2c[:, :3, :3], pts) + w2c[:, None, :3, 3]                     # [n_kf, n, 3]
        x, y, zc = -cam[..., 0], cam[..., 1], cam[..., 2]
        z = zc + 1e-5
        u, v = (fx * x + cx * zc) / z, (fy * y + cy * zc) / z
        edge = 10
        inside = (u < W - edge) & (u > edge) & (v < H - edge) & (v > edge) & (z < 0)
        percent = inside.float().mean(dim=1).cpu().numpy()
        self.last_overlap_percent = percent                      # inspection / tests
        order = sorted(range(len(keyframe_dict)), key=lambda i: percent[i], reverse=True)
        selected = [i for i in order if percent[i] > th]
        return [int(i) for i in np.random.permutation(np.array(selected))[:k]]

    # ------------------------------------------------------------------ slams/mapping.py:329-435
    def set_target_refer_frames(self, cur_gt_color, cur_gt_depth, cur_gt_label, cur_gt_c2w, cur_c2w):
        """Target / reference frame bundles of one ``optimize()`` call from ``self.keyframe_dict`` / ``self.keyframe_list``
        (filled by the caller's frame loop, reference ``Mapper.run`` :975,1085) -- the data layout of the reference's
        function, which is what the hot path consumes.  WHICH old keyframes join is host-side control flow outside the path
        (component 4): ``self.mapping_mode`` as in the reference -- 'overlap' = ``keyframe_selection_overlap``
        with th = 0.05 (:353-356), 'global' = k uniform draws with replacement (``random_select`` :160-168); the reference's
        ``run`` alternates the two between its outer joint iterations (:1031-1035), which is the caller's loop here.  A
        ``self.keyframe_selector(cur_color, cur_depth, cur_c2w, keyframe_dict[:-1], k)`` callable, if set, overrides both.
        Always joined: the latest keyframe and the current frame (index -1);
        keyframe 0 never is (:361).  Per target frame two reference keyframes (its neighbours) plus itself (:399-419)."""
        import numpy as np
        kfs, n_kf = self.keyframe_dict, len(self.keyframe_list)
        n_joint = self.cfg["mapping"]["n_joint_optimize_frames"]
        k = min(n_joint - 2, n_kf)
        picked = []
        if n_kf >= 2:
            if getattr(self, "keyframe_selector", None) is not None:
                picked = list(self.keyframe_selector(cur_gt_color, cur_gt_depth, cur_c2w, kfs[:-1], k))
            elif getattr(self, "mapping_mode", "global") == "overlap":
                picked = self.keyframe_selection_overlap(cur_gt_color, cur_gt_depth, cur_c2w, kfs[:-1], k, th=0.05)
            else:
                picked = [int(v) for v in np.random.choice(np.arange(n_kf - 1), size=k, replace=True)] if n_kf - 1 > 0 else [0]
            picked = sorted({int(i) for i in picked + [n_kf - 1]} - {0})
        target_idx = picked + [-1]
        self.n_target_frame = len(target_idx)
        dev = self.device
        cur = {"gt_color": cur_gt_color, "gt_depth": cur_gt_depth, "gt_label": cur_gt_label, "gt_c2w": cur_gt_c2w,
               "est_c2w": cur_c2w}
        tgt = {key: [] for key in cur}
        ref_color, ref_c2w, ref_idx, labels = [], [], [], []
        last = n_kf - 1
        for t in target_idx:
            src = cur if t == -1 else kfs[t]
            for key in tgt:
                v = src[key].to(dev)
                tgt[key].append(v if key == "gt_label" else v.float())
            labels.append(torch.unique(tgt["gt_label"][-1], sorted=True))
            if t == -1:
                pair = [max(last - 1, 0), max(last, 0)]
            elif t == last:
                pair = [max(last - 2, 0), max(last - 1, 0)]
            else:
                pair = [max(t - 1, 0), t + 1]
            ref_color.append(torch.stack([kfs[r]["gt_color"].to(dev).float() for r in pair] + [tgt["gt_color"][-1]], 0))
            ref_c2w.append(torch.stack([kfs[r]["est_c2w"].to(dev).float() for r in pair] + [tgt["est_c2w"][-1]], 0))
            ref_idx.append(pair + [-1])
        target_frames = {key: torch.stack(v, 0) for key, v in tgt.items()}
        target_frames["label_dict"] = torch.unique(torch.cat(labels, 0), sorted=True).cpu().int().numpy()
        target_frames["kf_idx"] = target_idx
        refer_frames = {"gt_color": torch.stack(ref_color, 0), "est_c2w": torch.stack(ref_c2w, 0), "kf_idx": ref_idx}
        return target_idx, target_frames, refer_frames

    # ------------------------------------------------------------------ slams/mapping.py:839-949
    def optimize(self, n_iters, cur_idx, cur_gt_color, cur_gt_depth=None, cur_gt_label=None, cur_gt_c2w=None, cur_c2w=None,
                 **kw):
        """The reference's signature and return value (slams/mapping.py:839): ``optimize(n_iters, idx, color, depth,
        label, gt_c2w, cur_c2w) -> (cur_c2w [4,4], fine_loss_dict)`` with keys loss_camera_tensor, p_loss, d_loss,
        l_loss, lt_loss, smooth_loss; refined keyframe poses are written back into ``self.keyframe_dict`` (:914-926).
        The frame bundles come from ``set_target_refer_frames``; the frozen image stem ``self.encoder`` (:846) runs once
        per call when set (else the 2-D code is zero).  ``optimize(n_iters, idx, target_frames_dict)`` -- the bundle given
        directly -- is accepted too and is what ``optimize_frames`` takes."""
        if isinstance(cur_gt_color, dict):
            return self.optimize_frames(n_iters, cur_idx, cur_gt_color, **kw)
        target_idx, target_frames, refer_frames = self.set_target_refer_frames(cur_gt_color, cur_gt_depth, cur_gt_label,
                                                                              cur_gt_c2w, cur_c2w)
        features = None
        if getattr(self, "encoder", None) is not None:
            features = self.encoder(refer_frames["gt_color"]).clone().detach()
        c2w, terms = self.optimize_frames(n_iters, cur_idx, target_frames, features=features, refer_frames=refer_frames, **kw)
        cam_err = 0.0
        cam7 = lambda m: torch.cat((get_quad_from_c2w(m), torch.as_tensor(m)[:3, 3].detach().float().cpu()), 0)
        if self.is_BA:
            for i in range(1, len(target_idx) - 1):
                kf = self.keyframe_dict[target_idx[i]]
                kf["est_c2w"] = target_frames["est_c2w"][i].clone()
                est = torch.cat((self.last_quad_list[i].detach(), self.last_T_list[i].detach()), 0).cpu()
                cam_err += float(torch.abs(cam7(kf["gt_c2w"]) - est).mean())
        est = torch.cat((self.last_quad_list[-1].detach(), self.last_T_list[-1].detach()), 0).cpu()
        cam_err = (cam_err + float(torch.abs(cam7(cur_gt_c2w) - est).mean())) / len(target_idx)
        out = {"loss_camera_tensor": cam_err}
        out.update({key: terms[key] for key in ("p_loss", "d_loss", "l_loss", "lt_loss", "smooth_loss") if key in terms})
        return c2w, out

    def optimize_frames(self, n_iters, cur_idx, target_frames, features=None, smooth=True, refer_frames=None):
        """Iteration driver on a given bundle: ``target_frames`` is what ``set_target_refer_frames`` returns
        (gt_color / gt_depth / gt_label per frame, est_c2w, label_dict, kf_idx); with stem ``features``
        [n_target, n_refer, C, h, w] and ``refer_frames`` the 2-D branch runs inside every iteration (:533-551)."""
        self.n_target_frame = len(target_frames["gt_color"])
        self.is_BA = cur_idx >= self.start_optimize_idx
        new_decoder_idx = self.set_decoder(target_frames)
        if len(new_decoder_idx) > 0 and cur_idx > 50:                   # :857-868 (first_frame_optimized is the caller's state)
            cur_class = set(torch.unique(target_frames["gt_label"][-1]).tolist())
            warm = [c for c in new_decoder_idx if float(c) in cur_class]
            new_decoder_idx = warm
            if warm:
                self.decoder_init(warm, target_frames["gt_color"][-1], target_frames["gt_depth"][-1],
                                  target_frames["gt_label"][-1], target_frames.get("gt_c2w", target_frames["est_c2w"])[-1],
                                  target_frames["est_c2w"][-1], smooth=smooth)
        optimizer, quad_list, T_list = self.set_optimizer(target_frames)
        optimizer.param_groups[0]["lr"] = self.lr
        optimizer.param_groups[1]["lr"] = self.BA_cam_lr * self.is_BA
        optimizer.param_groups[2]["lr"] = self.BA_cam_lr * self.is_BA
        prep = self.prepare_frames(target_frames)
        terms = {}
        lt_of = lambda it: (10 if it > n_iters // 2 else 0) if len(new_decoder_idx) > 0 else 10
        if getattr(self, "use_map_step", False) and self.static_shapes and self.fused_losses and \
                (features is None or features.dim() == 3 or refer_frames is not None):
            # the same iterations as the loop below as a fixed launch sequence over preallocated buffers (fused_step.MapStep:
            # no autograd graph, draws and routing prepared a step ahead on the side stream); Adam state lives in the MapStep
            from .fused_step import MapStep
            ms = MapStep(self, target_frames, quad_list, T_list, prep=prep, features=features, smooth=smooth,
                         refer_frames=refer_frames)
            for iter_ in range(n_iters):
                ms.set_lambda_lt(lt_of(iter_))
                ms.step(last=iter_ == n_iters - 1)
            ms.write_back()
            if n_iters > 0:
                _, terms = ms.losses()
            n_iters = 0
        for iter_ in range(n_iters):
            optimizer.zero_grad()
            samples = self.get_target_samples(target_frames, quad_list, T_list, refer_frames=refer_frames, features=features,
                                              prep=prep)
            loss, terms = self.iteration_loss(samples, lambda_lt=lt_of(iter_), smooth=smooth)
            loss.backward()
            optimizer.step()
        bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=self.device)
        pose = lambda k: torch.cat([torch.cat((get_rotation_from_quad(quad_list[k].detach()), T_list[k].detach()[:, None]), -1),
                                    bottom], dim=0)
        if self.is_BA:                                                  # :914-926: refined keyframe poses go back to the caller
            for i in range(1, self.n_target_frame - 1):
                target_frames["est_c2w"][i] = pose(i).to(target_frames["est_c2w"][i].device)
        cur_c2w = pose(self.n_target_frame - 1)
        target_frames["est_c2w"][-1] = cur_c2w.clone().to(target_frames["est_c2w"][-1].device)
        self.last_quad_list, self.last_T_list = quad_list, T_list
        return cur_c2w, terms
