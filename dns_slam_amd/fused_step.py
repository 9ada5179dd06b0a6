"""One mapping iteration as a fixed launch sequence (slams/mapping.py:870-912: sample -> render -> losses -> backward -> Adam).

``Mapper.optimize_frames`` drives the kernels through torch.autograd: correct for any caller, but the iteration is ~90
launches of which more than half are autograd's own glue (gradient accumulation, zero fills, ``stack`` / ``cat`` / slices,
one workspace allocation per backward node) and the host needs ~2.0 ms to enqueue a 2.3 ms iteration.  ``MapStep`` is the
same iteration written out by hand for the static-shape case (``mapper.static_shapes``): forward and backward are a fixed
list of C-ABI calls (include/dns_hip.h) over buffers allocated ONCE, every gradient lands in one flat buffer that a single
fill clears, the lattice (smoothness) branch -- forward AND backward, it meets the ray branch only in the parameter
gradients -- runs on the side stream, and Adam (csrc/adam.hip) steps all parameters from a prebuilt tensor list.

Everything that depends only on the random draws -- the draws themselves, the per-frame maximum of the sampled depths, the
class -> decoder routing of every sample (``dns_class_slots`` + ``dns_group_slots``) and the zero fills of the buffers the step
accumulates into -- is prepared ONE STEP AHEAD on the side stream into one of two alternating sets of buffers (``_Set``), so
none of it sits on the main stream's chain of dependent kernels.  The memory-bound kernels of the backward that nothing
downstream waits for -- ``dW_in`` of every network (``dns_mlp_dwin``), the pose gradient -- run on the side stream as well,
beside the vector-bound backward kernels / table scatter on the main stream (2.31 -> 1.90 ms per iteration
at 4096 rays x 64 samples).

Same kernels, same arithmetic, same random draws (``Mapper._draw_all``, same generator order) as the autograd path: tests/test_gpu_fused_step.py
holds the two against each other (losses, every parameter after several iterations).  No CPU path: the constructor raises
off-GPU like every op here.

Multi-GPU (``mapper.dist`` in weak mode, dist.py): the MAX of the per-frame sampled depth and the SUM of the 16 loss
numerators are all-reduced where the autograd path does it; the gradient buffer is laid out [colour | logit | fine pool |
table | coarse | poses] so that the first three -- complete when the ray branch's MLP backward ends -- travel (asynchronous
all-reduce) under the hash-grid scatter, and the rest after the two streams have joined.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import ops
from ._lib import DnsAdamTensor, DnsSplitRows, DnsTrackFused, check, ptr
from .common import get_quad_from_c2w, get_rotation_from_quad

_V = C.c_void_p


def _morton2(row, col):
    """Bit-interleaved (row, col) of int64 tensors below 2^16 each."""
    def spread(v):
        v = v & 0xffff
        v = (v | (v << 8)) & 0x00ff00ff
        v = (v | (v << 4)) & 0x0f0f0f0f
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
    return spread(col) | (spread(row) << 1)


class _Set:
    """What one step's preparation writes: the gradient buffer (zeroed), the fine network's output (zeroed), the draws and what
    derives from them alone (per-frame depth maximum, slot -> row table of the per-class routing)."""
    pass


class MapStep:
    def __init__(self, mapper, target_frames, quad_list=None, T_list=None, prep=None, features=None, lambda_lt=10.0,
                 smooth=True, betas=(0.9, 0.999), eps=1e-8, keep_hidden=False, split_rows=None, refer_frames=None):
        m = self.m = mapper
        dev = self.dev = torch.device(m.device)
        if dev.type != "cuda":
            raise ValueError("dns_slam_amd ops run on the GPU only; there is no CPU fallback")
        if not (m.static_shapes and m.fused_losses):
            raise ValueError("MapStep needs mapper.static_shapes and mapper.fused_losses (the sync-free iteration)")
        # features: None | the per-sample code [N, S, C] | the stem feature maps [n_target, n_refer, C, h, w] of the reference
        # views (+ refer_frames): then the 2-D branch of slams/mapping.py:532-557 -- feature_matching + Decoder.merge -- runs
        # INSIDE every step, forward and backward (Merge's weights train, its OneBlob input carries pose gradient)
        self.stem = features is not None and features.dim() == 5
        if features is not None and features.dim() not in (3, 5):
            raise ValueError("MapStep: features is the per-sample code [N, S, C] or the stem feature maps [K, R, C, h, w]")
        if self.stem and refer_frames is None:
            raise ValueError("MapStep: stem feature maps need refer_frames (kf_idx, est_c2w)")
        self.dist_on = m.dist is not None and m.dist.enabled       # (a forced one-rank group counts: tests drive RCCL that way)
        self.world = m.dist.world_size if self.dist_on else 1
        # union-batch mode (dist.py, SURVEY 8e's partitioning): every rank draws the SAME lists; this rank renders rays [a, b) of
        # every frame's list and x-planes [a_l, b_l) (+ one halo plane) of the ONE smoothness lattice
        self.union = self.dist_on and m.dist.union
        self.frames = target_frames
        self.prep = prep if prep is not None else m.prepare_frames(target_frames)
        self.npf_g = self.prep["n1"] + self.prep["n2"]                     # rays per frame of the whole (drawn) list
        self.ray_a, self.ray_b = m.dist.shard(self.npf_g) if self.union else (0, self.npf_g)
        self.features = None if (features is None or self.stem) else features.to(dev).float().contiguous()
        if self.features is not None and self.union:                        # the code of the rank's rays
            K_ = m.n_target_frame
            fs = self.features
            self.features = fs.reshape(K_, self.npf_g, *fs.shape[1:])[:, self.ray_a:self.ray_b].reshape(-1, *fs.shape[1:]).contiguous()
        self.lambda_lt, self.smooth = float(lambda_lt), bool(smooth)
        self.morton = (bool(getattr(m, "morton_draws", os.environ.get("DNS_MORTON_DRAWS", "0") == "1")) and m.label_layout == "per_ray")
        self.betas, self.eps = betas, eps
        K = self.K = m.n_target_frame
        dec, pool = m.decoder, m.fine_decoders
        self.pe_dim, self.grid_dim, self.hid = m.pe_dim, m.grid_dim, m.hidden_dim
        nets = (dec.coarse_fn.decoder, dec.out_fn.color_decoder, dec.out_fn.logit_decoder)
        if any(getattr(n, "fp16", False) != getattr(nets[0], "fp16", False) for n in nets):
            raise ValueError("MapStep: the networks must share one operand precision")
        self.fp16 = ops.MLP_FP16_FLAG if getattr(nets[0], "fp16", False) else 0
        shp = lambda n: (n.n_input_dims, n.n_output_dims, n.n_neurons, n.n_hidden_layers)
        self.shp_c, self.shp_col, self.shp_log = (shp(n) for n in nets)
        self.shp_f = (self.pe_dim + self.grid_dim, self.hid + 1, pool.nn_, pool.nl)
        # width of the colour / logit networks' second input segment: (fine latents | 2-D code).  WITHOUT a code (features None:
        # the reference multiplies a zero code through, slams/mapping.py:553-557) the code columns are identically zero: the
        # block is only the latents and the networks run with DNS_MLP_LIVE_IN -- their zero input columns are neither stored,
        # read nor multiplied (K 112 -> 80 in the first layer, no dX / dW_in work for them)
        code_dim = self.hid if self.stem else (0 if features is None else self.features.shape[-1])
        n_feat_full = self.shp_col[0] - self.pe_dim
        self.n_feat = self.hid + code_dim
        if not (self.pe_dim % 4 == 0 and self.pe_dim <= 64 and n_feat_full <= 64 and self.n_feat % 4 == 0
                and (self.n_feat == n_feat_full or code_dim == 0) and self.shp_log[0] == self.shp_col[0]):
            raise ValueError("MapStep: network shapes outside the two-segment input form (ops.render_nets has the same limits)")
        self.live = ops.MLP_LIVE_IN(self.pe_dim + self.n_feat) if self.n_feat < n_feat_full else 0
        if self.live and keep_hidden:                  # (the saved-activation backward takes full-width rows)
            self.n_feat, self.live = n_feat_full, 0
        self.n_class = self.shp_log[1]

        # ---- poses: one [K,4] / [K,3] pair is the parameter; frame 0 stays fixed when K > 1 (slams/mapping.py:447-455)
        if quad_list is None:
            quad_list = [get_quad_from_c2w(target_frames["est_c2w"][k]).to(dev) for k in range(K)]
            T_list = [target_frames["est_c2w"][k][:3, 3].detach().clone().to(dev) for k in range(K)]
        self.quad_list, self.T_list = quad_list, T_list
        self.Q = torch.stack([q.detach().float() for q in quad_list]).contiguous()
        self.T = torch.stack([t.detach().float() for t in T_list]).contiguous()
        self.is_BA = bool(m.is_BA)
        self.pose_row0 = 0 if K == 1 else 1

        # ---- parameters and the flat gradient buffer  [colour | logit | pool | table | coarse | quat | trans]
        self.p_table = dec.pe_fn.grid_fn.params
        self.meta = dec.pe_fn.grid_fn.meta
        self.n_bins = dec.pe_fn.pe_fn.n_bins
        self.p_coarse, self.p_color, self.p_logit = (n.params for n in nets)
        self.p_pool = pool.pool
        # [colour | logit | pool | merge (with stem features) || table | coarse | quat | trans]: what stands in front of the bar is
        # complete when the ray branch's MLP backward ends (the early all-reduce bucket under data parallelism)
        names = ["color", "logit", "pool"] + (["merge"] if self.stem else []) + ["table", "coarse", "quat", "trans"]
        if self.stem:
            mg = dec.merge.decoder
            self.shp_m = shp(mg)
            self.p_merge = mg.params
        byname = {"color": self.p_color, "logit": self.p_logit, "pool": self.p_pool, "table": self.p_table, "coarse": self.p_coarse,
                  "quat": self.Q, "trans": self.T}
        if self.stem:
            byname["merge"] = self.p_merge
        plist = [byname[n] for n in names]
        for p in plist:
            if not (p.is_cuda and p.is_contiguous() and p.dtype == torch.float32):
                raise ValueError("MapStep: parameters must be contiguous fp32 CUDA tensors")
        sizes = [(p.numel() + 3) // 4 * 4 for p in plist]                  # 16-byte aligned segments
        offs = [sum(sizes[:i]) for i in range(len(sizes))]
        off_of = dict(zip(names, offs))
        n_G = sum(sizes)
        self.M, self.V = torch.zeros(n_G, device=dev), torch.zeros(n_G, device=dev)
        self.adam_state = torch.zeros(3, device=dev)
        lr_pose = float(m.BA_cam_lr) * float(self.is_BA)
        items = [(byname[n], off_of[n], byname[n].numel(), float(m.lr)) for n in names if n not in ("quat", "trans")]
        if self.is_BA:
            a4, a3 = 4 * self.pose_row0, 3 * self.pose_row0
            items.append((self.Q.view(-1)[a4:], off_of["quat"] + a4, self.Q.numel() - a4, lr_pose))
            items.append((self.T.view(-1)[a3:], off_of["trans"] + a3, self.T.numel() - a3, lr_pose))
        items = [it for it in items if it[2] > 0]
        self.n_adam = len(items)

        def adam_items(G):
            arr = (DnsAdamTensor * len(items))()
            for k, (p, o, n, lr) in enumerate(items):
                it = arr[k]
                it.p, it.g = p.data_ptr(), G.data_ptr() + 4 * o
                it.m, it.v = self.M.data_ptr() + 4 * o, self.V.data_ptr() + 4 * o
                it.n, it.lr = n, lr
            return arr

        # ---- ray-branch buffers
        npf = self.npf = self.ray_b - self.ray_a
        nu = 0 if m.t_uniform is None else m.t_uniform.numel()
        ns = m.n_surface_ray
        N, S = K * npf, nu + ns
        P = N * S
        self.N, self.S, self.P, self.nu, self.ns = N, S, P, nu, ns
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        self.rays_o, self.rays_d, self.gt_color, self.gt_depth = f(N, 3), f(N, 3), f(N, 3), f(N)
        self.gt_label = torch.empty(N, device=dev, dtype=torch.int64)
        self.inside = torch.empty(N, device=dev, dtype=torch.uint8)
        self.z, self.pts = f(N, S), f(N, S, 3)
        ld = self.ld = self.pe_dim + self.grid_dim
        self.x3, self.buf = f(P, 3), f(P, ld)
        self.dydx = f(self.meta.n_levels * 3 * P * 2) if self.is_BA else None
        # Split rows (include/dns_hip.h, ABI v9): the encoder and the feature block write their rows ONCE in the form the MLP
        # kernels' matrix instructions take (f16 hi | lo halfs + one exponent per row) and every forward / backward launch loads
        # its operand fragments straight from memory; the fp32 rows are still written for the streaming dW_in kernels.
        # Measured (round 4): the MLP kernels do NOT get faster at fp32 grade (vector instructions -12 %,
        # time +-0: they are not bound by their input handling; cfg2 step 1.89 -> 2.04 ms) and with half-width networks the
        # forward's gain (0.60 -> 0.49 ms at cfg5_fp16) is eaten by the encoder writing a second row format (0.40 -> 0.51):
        # 6.19 -> 6.34 ms.  OFF by default; DNS_SPLIT_ROWS=1 / split_rows=True turns it on (tests/test_gpu_split_rows.py).
        if split_rows is None:
            split_rows = os.environ.get("DNS_SPLIT_ROWS", "0") == "1"
        self.sr = bool(split_rows) and ld % 16 == 0 and self.pe_dim % 16 == 0 and self.n_feat % 16 == 0 and not keep_hidden
        if self.sr and self.live:                      # the split-row entry points take full-width rows: keep the zero code columns
            self.n_feat, self.live = self.shp_col[0] - self.pe_dim, 0
        # ---- prepared operand images (round 5): every MLP launch of the step copies its weight images in (dns_mlp_prepare, built
        # once per step on the side stream under the ray branch's sampling / encoding) instead of building them per workgroup:
        # -10..17 us per backward launch and -2..6 us per forward launch since the copy-in keeps eight requests in flight.
        # mapper.prepared_images = False keeps the per-workgroup build; the split-row form takes fp32 weights
        # ---- HALF ROWS (ABI v12): networks that ask for tcnn's own precision (cfg['model']['mlp']['dtype'] = 'fp16', BASELINE
        # configs[4]) run on the native f16 kernels -- the encoder and the feature block write plain f16 rows, every network launch
        # reads them, ONE backward kernel per network forms all gradients (no dH_1 workspace, no dns_mlp_dwin), gradients carry
        # tcnn's static loss scale (mapper.loss_scale, default 128).  mapper.half_rows = False / DNS_HALF_ROWS=0 keeps round 4's
        # fp16-OPERAND mode of the split-operand kernels (per-point scales on fp32 rows).  The in-step 2-D branch (stem features)
        # and kept hidden activations are not wired to it.
        self.half = (bool(self.fp16) and bool(getattr(m, "half_rows", os.environ.get("DNS_HALF_ROWS", "1") != "0")) and not self.stem
                     and not keep_hidden and not self.sr and ld % 16 == 0 and self.pe_dim % 8 == 0 and self.n_feat % 8 == 0
                     and (self.pe_dim + self.n_feat) % 16 == 0)
        self.loss_scale = float(getattr(m, "loss_scale", ops.HALF_LOSS_SCALE))
        if self.half:
            self.xh = torch.empty(P, ld, device=dev, dtype=torch.float16)
            self.feath = torch.empty(P, self.n_feat, device=dev, dtype=torch.float16)
        self.use_prep = (bool(getattr(m, "prepared_images", os.environ.get("DNS_PREPARED_IMAGES", "1") != "0")) and not self.sr
                         and not self.half)
        self._blobs, self._prep_jobs = {}, []
        if self.use_prep:
            raw_lib = ops.lib._raw
            G = int(self.p_pool.shape[0]) if self.p_pool.dim() == 2 else 1
            live_n = (self.live >> 16) & 0xff
            jobs = [(self.p_coarse, self.shp_c, self.shp_c[1], 1, 0, 0),            # (params, shape, n_out, n_sets, stride, live flag)
                    (self.p_coarse, self.shp_c, 1, 1, 0, 0),                          # the lattice runs the occupancy row only
                    (self.p_pool, self.shp_f, self.shp_f[1], G, int(self.p_pool.shape[-1]), 0),
                    (self.p_color, self.shp_col, self.shp_col[1], 1, 0, self.live),
                    (self.p_logit, self.shp_log, self.shp_log[1], 1, 0, self.live)]
            for prm, shp_, n_out, n_sets, stride, lv in jobs:
                n_in_eff = live_n if lv else shp_[0]
                nfl = int(raw_lib.dns_mlp_prepared_floats(n_in_eff, n_out, shp_[2], shp_[3]))
                blob = torch.empty(nfl * n_sets + 64, device=dev)
                off = (-blob.data_ptr() // 4) % 4                      # 16-byte aligned start
                view = blob[off:off + nfl * n_sets]
                self._blobs[(prm.data_ptr(), n_out)] = view
                self._prep_jobs.append((prm, shp_[0], n_out, shp_[2], shp_[3], n_sets, stride, view, lv, blob))
        self.sr_planes = 1 if self.fp16 else 2           # half-width networks read the hi plane only
        self.sr_flags = 1 if self.fp16 else 0            # DNS_SPLIT_HI_ONLY
        if self.sr:
            np_ = self.sr_planes
            h16 = lambda *s_: torch.empty(*s_, device=dev, dtype=torch.float16)
            i32 = lambda n_: torch.empty(n_, device=dev, dtype=torch.int32)
            self.xs, self.xexp = h16(P, np_ * ld), i32(P)
            self.fxs, self.fexp = h16(P, np_ * self.n_feat), i32(P)
            self.rows_x = DnsSplitRows(self.xs.data_ptr(), self.xexp.data_ptr(), np_ * ld, ld if np_ == 2 else 0)
            self.rows_f = DnsSplitRows(self.fxs.data_ptr(), self.fexp.data_ptr(), np_ * self.n_feat, self.n_feat if np_ == 2 else 0)
        nf = self.hid + 1
        self.coarse = f(P, nf)
        n_groups = self.n_groups = max(len(pool), 1)
        self.n_slots = (P + 127) // 128 * 128 + 128 * n_groups
        self.group_ws = torch.empty(512, device=dev, dtype=torch.int32)
        self.feat = f(P, self.n_feat)
        self.slot = torch.empty(P, device=dev, dtype=torch.int64)
        # two alternating sets of everything a step's preparation writes (see _prepare)
        self.sets = []
        for _ in range(2):
            st_ = _Set()
            o_G = (P * nf + 3) // 4 * 4                                   # G starts 16-byte aligned
            st_.zb = torch.zeros(o_G + n_G, device=dev)                   # [fine | G]: one fill per step clears both
            st_.fine, st_.G = st_.zb[:P * nf].view(P, nf), st_.zb[o_G:]
            for n_ in names:
                setattr(st_, "g_" + n_, st_.G[off_of[n_]:off_of[n_] + byname[n_].numel()])
            st_.G_early, st_.G_late = st_.G[:off_of["table"]], st_.G[off_of["table"]:]
            st_.adam_items = adam_items(st_.G)
            st_.row_index = torch.empty(self.n_slots, device=dev, dtype=torch.int32)
            st_.tile_group = torch.empty(self.n_slots // 128, device=dev, dtype=torch.int32)
            st_.ev, st_.draws, st_.dmax = None, None, None
            self.sets.append(st_)
        self.cur = self.sets[0]
        self.raw, self.logit = f(P, 4), f(P, self.n_class)
        self.depth, self.var, self.rgb, self.weights, self.sem = f(N), f(N), f(N, 3), f(N, S), f(N, self.n_class)
        self.sums_ws, self.out = f(ops.LOSS_SUMS_FLOATS), f(16)
        self.one = torch.ones(1, device=dev)
        self.d_color, self.d_depth, self.d_sem = f(N, 3), f(N), f(N, self.n_class)
        self.d_coarse = f(P, nf)
        self.d_raw, self.d_logit, self.d_col = f(P, 4), f(P, self.n_class), f(P, 4)
        self.d_buf = f(P, ld)
        self.d_featx = torch.zeros(P, 4 + self.n_feat, device=dev)        # zeroed once: the code columns only accumulate
        self.d_x3 = f(P, 3)
        raw_lib = ops.lib._raw
        if self.stem:
            Kf, R, Cs, fh, fw = features.shape
            if Kf != K:
                raise ValueError("MapStep: stem feature maps must have one set of reference views per target frame")
            self.R, self.Cs, self.fh, self.fw = int(R), int(Cs), int(fh), int(fw)
            self.feat_maps = features.detach().to(dev).float().permute(0, 1, 3, 4, 2).contiguous().reshape(K * R, fh, fw, Cs)
            mrg = dec.merge
            self.n_bins_m = mrg.pe_fn.n_bins
            self.pe_m = 3 * self.n_bins_m
            n_in_m, n_out_m = self.shp_m[0], self.shp_m[1]
            if not (n_in_m == self.pe_m + Cs and n_out_m == self.hid and self.pe_m % 4 == 0):
                raise ValueError("MapStep: Decoder.merge's network does not match the stem features / hidden width")
            self.b6m = ops._bound6(mrg.bound)
            # which pose every reference view takes (slams/mapping.py:534-545): -1 = the target frame itself, a keyframe that is
            # also a target = that target's pose under optimisation, anything else = the stored keyframe pose
            target_idx = list(target_frames["kf_idx"])
            src, fixed = [], []
            for i in range(K):
                ridx = list(refer_frames["kf_idx"][i])
                if len(ridx) != R:
                    raise ValueError("MapStep: refer_frames['kf_idx'] must list n_refer views per target frame")
                for r, rid in enumerate(ridx):
                    k = i if rid == -1 else (target_idx.index(rid) if rid in target_idx else -1)
                    src.append(k)
                    fixed.append(refer_frames["est_c2w"][i][ridx.index(rid)].detach().to(dev).float().reshape(16))
            self.ref_src = torch.tensor(src, dtype=torch.int32, device=dev)
            self.ref_fixed = torch.stack(fixed).contiguous()
            self.w2c, self.origin = f(K * R, 16), f(K * R, 3)
            Mr = self.Mr = R * P
            self.Pf = npf * S                                             # points per target frame
            self.mbuf, self.rel, self.xm = f(Mr, n_in_m), f(Mr, 3), f(Mr, 3)
            self.mlat, self.mdy = f(Mr, n_out_m), f(Mr, n_out_m)
            self.d_mpe, self.d_rel = (f(Mr, self.pe_m), f(Mr, 3)) if self.is_BA else (None, None)
            self.ws_m = f(max(int(raw_lib.dns_mlp_bwd_ws_floats(Mr, self.shp_m[2], self.shp_m[3])), 4))
            self.K9 = (C.c_float * 9)(float(m.fx), 0.0, float(m.cx), 0.0, float(m.fy), float(m.cy), 0.0, 0.0, 1.0)
        self.ray_ws = f(max(int(raw_lib.dns_raygen_bwd_ws_floats(K, npf)), 1))
        mlp_ws = lambda n_slots, s: f(max(int(raw_lib.dns_mlp_bwd_ws_floats(n_slots, s[2], s[3])), 4))
        self.ws_mlp = max((mlp_ws(self.n_slots, s) for s in (self.shp_f, self.shp_c, self.shp_col, self.shp_log)),
                          key=lambda t: t.numel())
        self.ws_mlp4 = [self.ws_mlp] + [torch.empty_like(self.ws_mlp) for _ in range(3)]   # one per network when dW_in is forked
        # Hidden activations kept by the forward for the backward (dns_mlp_fwd h_save -> dns_mlp_bwd h_saved) instead of being
        # recomputed: stand-alone the backward kernel is 17 % faster (122 -> 101 us) and the forward 18 % slower (46 -> 54 us),
        # but in this two-stream step the extra 0.76 GB of traffic per iteration costs more than the vector work it saves
        # (1.917 -> 1.944 ms with every network keeping them, 1.915 with the lattice branch alone): off by default
        self.keep_h = self.keep_h_lat = bool(keep_hidden)
        hbuf = lambda n_slots, s: f(s[3] * n_slots * s[2]) if self.keep_h else None
        self.h_c, self.h_f, self.h_col, self.h_log = hbuf(P, self.shp_c), hbuf(self.n_slots, self.shp_f), hbuf(P, self.shp_col), hbuf(P, self.shp_log)
        self.scatter_form, self.scatter_cap = ops.SCATTER_FORM
        enc_ws = lambda n: f(max(int(raw_lib.dns_encode_bwd_ws_floats(n, C.byref(self.meta.c), self.scatter_form, self.scatter_cap)), 4))
        self.ws_enc = enc_ws(P)
        tr = m.cfg["training"]
        self.lam = (C.c_float * 8)(m.lambda_p, m.lambda_d, m.lambda_l, self.lambda_lt, m.lambda_fs, m.lambda_opacity,
                                   tr["opacity_sigma"], 0.05)
        self.camv = (C.c_double * 4)(float(m.fx), float(m.fy), float(m.cx), float(m.cy))
        self.b6 = ops._bound6(m.bound)

        # ---- lattice branch (slams/mapping.py:129-159)
        if self.smooth:
            sp = self.sp = tr["smooth_pts"]
            m._ensure_lattice(sp, 0.1, 0.05)
            n = self.n_lat = sp - 1
            la, lb = m.dist.shard(n) if self.union else (0, n)
            lhi = min(lb + 1, n)                                          # + the next slab's first plane (halo: closes the x-differences)
            self.lat_nx, self.lat_halo = lhi - la, 1 if lhi > lb else 0
            Pl = self.Pl = self.lat_nx * n * n
            self.bufl, self.occ, self.d_occ, self.d_bufl = f(Pl, ld), f(Pl, 1), f(Pl, 1), f(Pl, ld)
            if self.half:
                self.xhl = torch.empty(Pl, ld, device=dev, dtype=torch.float16)
            if self.sr:
                np_ = self.sr_planes
                self.xsl, self.xexpl = torch.empty(Pl, np_ * ld, device=dev, dtype=torch.float16), torch.empty(Pl, device=dev, dtype=torch.int32)
                self.rows_l = DnsSplitRows(self.xsl.data_ptr(), self.xexpl.data_ptr(), np_ * ld, ld if np_ == 2 else 0)
            self.pts_l = f(Pl, 3)
            # Row m of the lattice branch's buffers holds lattice element lat_order[m], the elements in MORTON order of (i, j, k):
            # neighbouring rows then share hash-table lines and the lattice's encoding kernel runs 2.3x faster (112 -> 49 us at
            # 63^3; x-major rows walk the z axis and touch new lines of every fine level at every step).  The scatter does not
            # care (249 -> 256 us).  lat_slot = the inverse map (row of element e).
            ar = torch.arange(n, device=dev)
            ii, jj, kk = torch.meshgrid(ar[la:lhi], ar, ar, indexing="ij")      # the rank's x-planes (the whole cube without union mode)
            elem = (ii * (n * n) + jj * n + kk).reshape(-1)                      # x-major index of every element of the slab

            def spread(v):                          # bits of v to every third position (10 bits)
                v = v & 0x3ff
                v = (v | (v << 16)) & 0x30000ff
                v = (v | (v << 8)) & 0x300f00f
                v = (v | (v << 4)) & 0x30c30c3
                return (v | (v << 2)) & 0x9249249

            code = spread(ii.reshape(-1)) | (spread(jj.reshape(-1)) << 1) | (spread(kk.reshape(-1)) << 2)
            self.lat_order_l = torch.argsort(code).contiguous()                 # int64, for index_select: row m = slab element [m]
            self.lat_order = elem[self.lat_order_l].to(torch.int32).contiguous()  # ... as x-major index in the CUBE (the kernel's input)
            self.lat_slot = torch.empty_like(self.lat_order_l)
            self.lat_slot[self.lat_order_l] = torch.arange(Pl, device=dev)
            self.occ_x, self.d_occ_x = f(Pl, 1), f(Pl, 1)
            _, c_vox, c_off, c_mar = m._lattice_consts                    # float64 [3] each, on the device: read once
            self.lat_consts = (C.c_double * 9)(*[float(v) for t in (c_vox, c_off, c_mar) for v in t.cpu().tolist()])
            self.tv = f(1)
            # gradients are SUMMED over the ranks: weak mode averages the ranks' own lattices (1 / world each), union mode adds
            # up the slabs of one lattice
            self.w_sm = torch.full((1,), m.lambda_sm if self.union else m.lambda_sm / self.world, device=dev)
            self.ws_mlp_l = mlp_ws(Pl, self.shp_c)
            self.h_l = f(self.shp_c[3] * Pl * self.shp_c[2]) if self.keep_h_lat else None
            self.ws_enc_l = enc_ws(Pl)
        if getattr(m, "_side_stream", None) is None:
            m._side_stream = torch.cuda.Stream(device=dev)
        self.side = m._side_stream
        self.steps = 0

    # ------------------------------------------------------------------------------------------------------------------
    def _prepare_weights(self, st):
        """The step's operand images from the CURRENT weights (after the last Adam step): five small launches."""
        for prm, n_in, n_out, nn, nl, n_sets, stride, view, lv, _ in self._prep_jobs:
            check(ops.lib.dns_mlp_prepare(ptr(prm), n_in, n_out, nn, nl, n_sets, stride, ptr(view), lv, st), "dns_mlp_prepare")

    def _w(self, params, n_out):
        """(pointer, flag) of a network's weights for an MLP launch: its prepared images, or the fp32 parameters."""
        blob = self._blobs.get((params.data_ptr(), n_out)) if self.use_prep else None
        if blob is None:
            return ptr(params), 0
        return ptr(blob), ops.MLP_PREPARED_FLAG

    def _lattice_branch(self, cur, st):
        """Forward and backward of the smoothness term on the side stream: its loss weight is a constant, so its backward
        needs nothing from the ray branch."""
        lib = ops.lib
        Pl, ld, pe = self.Pl, self.ld, self.pe_dim
        pts = self.pts_l
        check(lib.dns_lattice_points(ptr(cur.draws["r6"]), self.lat_consts, self.n_lat, ptr(self.lat_order), Pl, ptr(pts), st),
              "dns_lattice_points")
        meta = C.byref(self.meta.c)
        grid_l = _V(self.bufl.data_ptr() + 4 * pe)
        n_in, _, nn, nl = self.shp_c
        if self.half:
            check(lib.dns_encode_fwd_split(ptr(pts), None, Pl, self.n_bins, ptr(self.p_table), meta, None, None, 0, ptr(self.xhl), ld,
                                           None, ops.SPLIT_PLAIN, None, st), "dns_encode_fwd_split")
            check(lib.dns_mlp_fwd_half(ptr(self.xhl), ld, None, 0, 0, ptr(self.p_coarse), n_in, 1, nn, nl, ptr(self.occ), 1, Pl,
                                       None, None, 0, 0, st), "dns_mlp_fwd_half")
        elif self.sr:
            check(lib.dns_encode_fwd_split(ptr(pts), None, Pl, self.n_bins, ptr(self.p_table), meta, None, ptr(self.bufl), ld,
                                           ptr(self.xsl), self.sr_planes * ld, ptr(self.xexpl), self.sr_flags, None, st), "dns_encode_fwd_split")
            check(lib.dns_mlp_fwd_split(C.byref(self.rows_l), None, 0, ptr(self.p_coarse), n_in, 1, nn, nl, ptr(self.occ), 1, Pl,
                                        None, None, 0, self.fp16, st), "dns_mlp_fwd_split")
        else:
            check(lib.dns_encode_fwd(ptr(pts), None, Pl, self.n_bins, ptr(self.p_table), meta, None, ptr(self.bufl), ld,
                                     grid_l, ld, None, st), "dns_encode_fwd")
            wl, wflag = self._w(self.p_coarse, 1)
            check(lib.dns_mlp_fwd(ptr(self.bufl), ld, None, 0, 0, wl, n_in, 1, nn, nl, ptr(self.occ), 1, Pl,
                                  None, None, 0, ptr(self.h_l), self.fp16 | wflag, st), "dns_mlp_fwd")
        # the branch runs in MORTON order of the lattice elements (see __init__); the TV kernels want the x-major cube: two 1 MB
        # permutations (the network's occupancy out, its gradient back in)
        torch.index_select(self.occ, 0, self.lat_slot, out=self.occ_x)
        check(lib.dns_tv_fwd(ptr(self.occ_x), 1, self.lat_nx, self.n_lat, self.lat_halo, self.sp, ptr(self.tv), st), "dns_tv_fwd")
        check(lib.dns_tv_bwd(ptr(self.occ_x), 1, self.lat_nx, self.n_lat, self.lat_halo, self.sp, ptr(self.w_sm), ptr(self.d_occ_x), st),
              "dns_tv_bwd")
        torch.index_select(self.d_occ_x, 0, self.lat_order_l, out=self.d_occ)
        if self.half:
            # (the lattice points carry no pose: only the grid columns' input gradient has a consumer, the table scatter)
            check(lib.dns_mlp_bwd_half(ptr(self.xhl), ld, None, 0, 0, ptr(self.d_occ), 1, ptr(self.p_coarse), n_in, 1, nn, nl,
                                       ptr(self.d_bufl), ld, None, 0, ptr(cur.g_coarse), Pl, None, None, 0, ops.MLP_DX_FROM(pe),
                                       self.loss_scale, st), "dns_mlp_bwd_half")
        elif self.sr:
            check(lib.dns_mlp_bwd_split(C.byref(self.rows_l), None, 0, ptr(self.d_occ), 1, ptr(self.p_coarse), n_in, 1, nn, nl,
                                        ptr(self.d_bufl), ld, None, 0, ptr(cur.g_coarse), ptr(self.ws_mlp_l), Pl, None, None, 0,
                                        self.fp16, st), "dns_mlp_bwd_split")
            check(lib.dns_mlp_dwin(ptr(self.bufl), ld, None, 0, 0, n_in, nn, nl, ptr(cur.g_coarse), ptr(self.ws_mlp_l), Pl, None, None,
                                   0, self.fp16, st), "dns_mlp_dwin")
        else:
            # (the lattice points carry no pose: only the grid columns' input gradient has a consumer, the table scatter)
            wl, wflag = self._w(self.p_coarse, 1)
            check(lib.dns_mlp_bwd(ptr(self.bufl), ld, None, 0, 0, ptr(self.d_occ), 1, wl, n_in, 1, nn, nl,
                                  ptr(self.d_bufl), ld, None, 0, ptr(cur.g_coarse), ptr(self.ws_mlp_l), Pl, None, None, 0, ptr(self.h_l),
                                  self.fp16 | wflag | (0 if self.h_l is not None else ops.MLP_DX_FROM(pe)), st), "dns_mlp_bwd")
        d_grid_l = _V(self.d_bufl.data_ptr() + 4 * pe)
        check(lib.dns_encode_bwd(ptr(pts), None, Pl, self.n_bins, ptr(self.p_table), meta, None, ld, d_grid_l, ld,
                                 ptr(cur.g_table), None, None, ptr(self.ws_enc_l), self.scatter_form, self.scatter_cap, st),
              "dns_encode_bwd")

    def _prepare(self, st_, stream, draws):
        """Everything of a step that depends on the random draws alone, on the CURRENT stream (``stream`` = its handle): the
        draws (utils/common.py:274,313-328,571-582; slams/mapping.py:137-143 -- ``Mapper._draw_all``, the generator order of the
        autograd path), the per-frame maximum of the sampled depths (:581,591; all-reduced MAX over the ranks), the routing of
        every sample to its class's decoder (slams/mapping.py:590-601,613) and the zero fills of what the step adds into."""
        m, lib = self.m, ops.lib
        K, npf, N, S, P = self.K, self.npf, self.N, self.S, self.P
        npf_g = self.npf_g                                              # the drawn list (all ranks' rays in union mode)
        prep = self.prep
        if draws is None:
            # Mapper._draw_all's generator calls in its order (pixels: randint then float64 rand; jitter; lattice), the index
            # arithmetic, the labels of the drawn pixels and the per-frame depth maxima in ONE kernel (dns_draw_finish)
            n1, n2 = prep["n1"], prep["n2"]
            i1 = torch.randint(prep["HW"], (K, n1), device=self.dev)
            u = torch.rand(K, n2, device=self.dev, dtype=torch.float64)
            pix = torch.empty(K * npf_g, device=self.dev, dtype=torch.int64)
            labels = torch.empty(K * npf_g, device=self.dev, dtype=torch.int64)
            dmax = torch.empty(K, device=self.dev, dtype=torch.int32)
            check(lib.dns_draw_finish(ptr(i1), ptr(u), ptr(prep["counts_f64"]), ptr(prep["counts_m1"]), ptr(prep["starts_flat"]),
                                      ptr(prep["sorted_flat"]), ptr(prep["depth"]), ptr(prep["label"]), K, n1, n2, prep["HW"],
                                      ptr(pix), ptr(labels), ptr(dmax), stream), "dns_draw_finish")
            if self.morton:
                # VERDICT r4 item 7 (an option, off by default; label_layout "per_ray" only -- the reference-tiled layout routes point k
                # by labels[k mod N], a function of the ray ORDER): every frame's drawn pixels in Morton order of (row, col), so that
                # neighbouring rays -- and with them neighbouring samples -- share hash-table lines.  The draw is the same SET; three torch
                # launches per step on the stream that prepares the next step's set (off the critical path with prefetch_draws).
                p2 = pix.view(K, npf_g)
                code = _morton2(torch.div(p2, m.W, rounding_mode="floor"), p2 % m.W)
                order = torch.argsort(code, dim=1)
                pix = torch.gather(p2, 1, order).reshape(-1).contiguous()
                labels = torch.gather(labels.view(K, npf_g), 1, order).reshape(-1).contiguous()
            d = {"pix": pix, "jitter": m.draw_jitter(), "r6": torch.rand(6, device=self.dev) if self.smooth else None}
        else:                                          # given draws (tests): the same quantities with torch ops
            d = dict(draws)                                # ('pix' = the WHOLE drawn list in union mode, like the generator's)
            pix2 = d["pix"].reshape(K, npf_g)
            dmax = torch.gather(prep["depth"].reshape(K, -1), 1, pix2).amax(dim=1).clamp_min(0.0).float().contiguous().view(torch.int32)
            labels = torch.gather(prep["label"].reshape(K, -1), 1, pix2).reshape(-1).long()      # = the gt_label raygen writes
        tiled = m.label_layout == "reference_tiled"
        n_lab, s_lab = N, S
        if self.union:
            # every rank holds the whole list: the per-frame depth maximum (utils/common.py:581,591) needs no collective; the rank
            # renders rays [a, b) of every frame; under the reference's tiled label layout (slams/mapping.py:613) point
            # k of the WHOLE batch is routed by labels[k mod N_whole] -- a function of the global index (dist.union_point_labels)
            m._check_same_draws(d["pix"].reshape(K, npf_g))
            a, b = self.ray_a, self.ray_b
            d["pix"] = d["pix"].reshape(K, npf_g)[:, a:b].reshape(-1).contiguous()
            if tiled:
                from .dist import union_point_labels
                labels, n_lab, s_lab, tiled = union_point_labels(labels, K, npf_g, a, b, S).contiguous(), P, 1, False
            else:
                labels = labels.reshape(K, npf_g)[:, a:b].reshape(-1).contiguous()
        st_.draws = d
        if self.dist_on and not self.union:
            fmax = dmax.view(torch.float32)
            m.dist.allreduce_max(fmax)                  # in place on the same storage
        st_.dmax = dmax                                 # bit patterns, as dns_raygen_sample takes them
        lut = m.fine_decoders.lut(0)                                       # class id -> pool row; :613 tiles the labels
        check(lib.dns_class_slots(ptr(labels), n_lab, s_lab, 1 if tiled else 0, ptr(lut), lut.numel(),
                                  ptr(self.slot), stream), "dns_class_slots")
        check(lib.dns_group_slots(ptr(self.slot), P, self.n_groups, 2, self.n_slots, ptr(self.group_ws), ptr(st_.row_index),
                                  ptr(st_.tile_group), stream), "dns_group_slots")
        st_.zb.zero_()            # the fine network's output (points without a network keep zeros, :592) | the gradient buffer

    @torch.no_grad()
    def step(self, draws=None, last=False):
        """One iteration; returns nothing (``losses()`` reads the terms of the last step).  ``draws``: {'pix': [K * n_per_frame]
        int64 pixel indices, 'jitter': (t_surf, t_zero), 'r6': [6] lattice offset | jitter} instead of the generator's (tests).
        ``last``: no further step follows -- the NEXT step's set is not prepared (with ``mapper.prefetch_draws`` every step
        otherwise consumes the generator's draws of its successor: after the final step of an optimize() the generator would
        stand one iteration further than the autograd driver leaves it)."""
        m, lib = self.m, ops.lib
        if torch.cuda.is_current_stream_capturing():
            self._captured = True                      # the loss weights are launch ARGUMENTS: a replay keeps them (set_lambda_lt)
        main = torch.cuda.current_stream()
        st = _V(main.cuda_stream)
        K, npf, N, S, P, ld, pe = self.K, self.npf, self.N, self.S, self.P, self.ld, self.pe_dim
        # mapper.prefetch_draws: this step's set was prepared on the side stream a step ago and the next one is prepared there
        # now; mapper.overlap_smooth: the lattice branch runs on the side stream (both off inside a one-stream graph capture)
        prefetch = getattr(m, "prefetch_draws", False) and draws is None
        cur = self.sets[self.steps % 2]
        if not prefetch or cur.ev is None:
            self._prepare(cur, st, draws)              # first step / no prefetch / given draws: in line, on the main stream
        else:
            main.wait_event(cur.ev)
        cur.ev = None
        self.cur = cur
        pix, (t_surf, t_zero) = cur.draws["pix"], cur.draws["jitter"]
        on_side = self.smooth and getattr(m, "overlap_smooth", False)
        if on_side or prefetch:
            self.side.wait_stream(main)                # behind the last Adam step and the last reads of the other set's buffers
        prep_ev = None
        if self.use_prep:
            if on_side or prefetch:                    # under the ray branch's sampling and encoding; the lattice branch follows it
                with torch.cuda.stream(self.side):
                    self._prepare_weights(_V(self.side.cuda_stream))
                    prep_ev = torch.cuda.Event()
                    prep_ev.record(self.side)
            else:
                self._prepare_weights(st)
        # (host order: the side stream's launches go out first.  Measured against "main stream's first launches first" and
        # "next set's preparation after the forward": 2.06 / 2.08 / 2.09 ms per step at cfg2 -- the lattice branch and the ~20
        # tiny preparation kernels overlap best with the ray branch's long FORWARD kernels, not with its backward.)
        if on_side:
            with torch.cuda.stream(self.side):
                self._lattice_branch(cur, _V(self.side.cuda_stream))
        elif self.smooth:
            self._lattice_branch(cur, st)
        if prefetch and not last:                      # the next step's set, behind the lattice branch on the side stream
            nxt = self.sets[(self.steps + 1) % 2]
            with torch.cuda.stream(self.side):
                self._prepare(nxt, _V(self.side.cuda_stream), None)
                nxt.ev = torch.cuda.Event()
                nxt.ev.record(self.side)
            for t in (nxt.draws["pix"], nxt.draws["jitter"][0], nxt.draws["jitter"][1], nxt.dmax):
                t.record_stream(main)                  # allocated on the side stream, read by the main stream next step

        # ---- rays, samples (utils/common.py:248-264, 561-599)
        jstride = self.ns if t_surf.dim() == 2 else 0
        prep = self.prep
        H, W = m.H, m.W
        check(lib.dns_raygen_sample(ptr(pix), ptr(prep["color"]), ptr(prep["depth"]), ptr(prep["label"]), ptr(self.Q),
                                    ptr(self.T), self.camv, self.b6, H, W, 0, H, 0, W, K, npf, ptr(m.t_uniform), ptr(t_surf),
                                    ptr(t_zero), self.nu, self.ns, jstride, ptr(cur.dmax), 1, ptr(self.rays_o), ptr(self.rays_d),
                                    ptr(self.gt_color), ptr(self.gt_depth), ptr(self.gt_label), ptr(self.inside), ptr(self.z),
                                    ptr(self.pts), st), "dns_raygen_sample")
        # ---- encoding (slams/mapping.py:608 + models/decoder.py:45-48)
        meta = C.byref(self.meta.c)
        grid = _V(self.buf.data_ptr() + 4 * pe)
        sr = self.sr
        half = self.half
        if half:
            check(lib.dns_encode_fwd_split(ptr(self.pts), self.b6, P, self.n_bins, ptr(self.p_table), meta, ptr(self.x3), None, 0,
                                           ptr(self.xh), ld, None, ops.SPLIT_PLAIN, ptr(self.dydx), st), "dns_encode_fwd_split")
        elif sr:
            check(lib.dns_encode_fwd_split(ptr(self.pts), self.b6, P, self.n_bins, ptr(self.p_table), meta, ptr(self.x3), ptr(self.buf),
                                           ld, ptr(self.xs), self.sr_planes * ld, ptr(self.xexp), self.sr_flags, ptr(self.dydx), st),
                  "dns_encode_fwd_split")
        else:
            check(lib.dns_encode_fwd(ptr(self.pts), self.b6, P, self.n_bins, ptr(self.p_table), meta, ptr(self.x3), ptr(self.buf),
                                     ld, grid, ld, ptr(self.dydx), st), "dns_encode_fwd")
        # ---- the 2-D branch inside the iteration (slams/mapping.py:532-551): reference poses -> projected image code + relative
        # point -> OneBlob -> Merge network; its mean over the views is taken by the feature block below
        if self.stem:
            R, Pf = self.R, self.Pf
            n_in_m, n_out_m, nn_m, nl_m = self.shp_m
            check(lib.dns_refer_poses(ptr(self.Q), ptr(self.T), ptr(self.ref_src), ptr(self.ref_fixed), K * R, ptr(self.w2c),
                                      ptr(self.origin), st), "dns_refer_poses")
            check(lib.dns_feature_gather_frames(ptr(self.pts), ptr(self.w2c), ptr(self.origin), self.K9, ptr(self.feat_maps), K, R, Pf,
                                                self.Cs, self.fh, self.fw, H, W, _V(self.mbuf.data_ptr() + 4 * self.pe_m), n_in_m,
                                                ptr(self.rel), st), "dns_feature_gather_frames")
            check(lib.dns_encode_fwd(ptr(self.rel), self.b6m, self.Mr, self.n_bins_m, None, None, ptr(self.xm), ptr(self.mbuf), n_in_m,
                                     None, 0, None, st), "dns_encode_fwd")
            check(lib.dns_mlp_fwd(ptr(self.mbuf), n_in_m, None, 0, 0, ptr(self.p_merge), n_in_m, n_out_m, nn_m, nl_m, ptr(self.mlat),
                                  n_out_m, self.Mr, None, None, 0, None, self.fp16, st), "dns_mlp_fwd")
        # ---- the four networks (slams/mapping.py:616-626)
        fp16 = self.fp16
        rows_x = C.byref(self.rows_x) if sr else None
        rows_f = C.byref(self.rows_f) if sr else None

        def fwd(x2, n_in1, params, shape, y, ri, tg, n_slots, stride, hs):
            n_in, n_out, nn, nl = shape
            if half:                                   # x2: the f16 feature block (self.feath) where the fp32 paths take self.feat
                check(lib.dns_mlp_fwd_half(ptr(self.xh), ld, None if x2 is None else ptr(self.feath), self.n_feat, n_in1, ptr(params),
                                           n_in, n_out, nn, nl, ptr(y), y.stride(0), n_slots, ptr(ri), ptr(tg), stride,
                                           0 if x2 is None else self.live, st), "dns_mlp_fwd_half")
                return
            if sr:
                check(lib.dns_mlp_fwd_split(rows_x, None if x2 is None else rows_f, n_in1, ptr(params), n_in, n_out, nn, nl, ptr(y),
                                            y.stride(0), n_slots, ptr(ri), ptr(tg), stride, fp16, st), "dns_mlp_fwd_split")
                return
            wp, wflag = self._w(params, n_out)
            check(lib.dns_mlp_fwd(ptr(self.buf), ld, ptr(x2), 0 if x2 is None else x2.stride(0), n_in1, wp, n_in, n_out,
                                  nn, nl, ptr(y), y.stride(0), n_slots, ptr(ri), ptr(tg), stride, ptr(hs),
                                  fp16 | wflag | (0 if x2 is None else self.live), st), "dns_mlp_fwd")

        if prep_ev is not None:
            main.wait_event(prep_ev)                   # the operand images of this step (side stream)
        fwd(None, 0, self.p_coarse, self.shp_c, self.coarse, None, None, P, 0, self.h_c)
        fine, row_index, tile_group = cur.fine, cur.row_index, cur.tile_group     # zeroed / routed by _prepare
        fwd(None, 0, self.p_pool, self.shp_f, fine, row_index, tile_group, self.n_slots, self.p_pool.shape[-1], self.h_f)
        # (latents | truncated 2-D code) for the colour / logit networks, occupancy into the compositing input (:553-556, :622-627)
        if half:
            check(lib.dns_feature_block_split(ptr(fine), self.hid + 1, self.hid, ptr(self.features), self.n_feat - self.hid, 1, 0,
                                              ptr(self.z), ptr(self.gt_depth), N, S, None, 0, ptr(self.feath), self.n_feat, None,
                                              ops.SPLIT_PLAIN, ptr(self.raw), st), "dns_feature_block_split")
        elif sr or self.stem:
            code, n_ref, Pf_ = (self.mlat, self.R, self.Pf) if self.stem else (self.features, 1, 0)
            check(lib.dns_feature_block_split(ptr(fine), self.hid + 1, self.hid, ptr(code), self.n_feat - self.hid, n_ref, Pf_,
                                              ptr(self.z), ptr(self.gt_depth), N, S, ptr(self.feat), self.n_feat,
                                              ptr(self.fxs) if sr else None, self.sr_planes * self.n_feat,
                                              ptr(self.fexp) if sr else None, self.sr_flags, ptr(self.raw), st),
                  "dns_feature_block_split")
        else:
            check(lib.dns_feature_block(ptr(fine), self.hid + 1, self.hid, ptr(self.features), self.n_feat - self.hid, ptr(self.z),
                                        ptr(self.gt_depth), N, S, ptr(self.feat), self.n_feat, ptr(self.raw), st), "dns_feature_block")
        fwd(self.feat, pe, self.p_color, self.shp_col, self.raw, None, None, P, 0, self.h_col)
        fwd(self.feat, pe, self.p_logit, self.shp_log, self.logit, None, None, P, 0, self.h_log)
        # ---- compositing + losses (utils/common.py:506-537, slams/mapping.py:887-907).  raw[:, 0:3] stays the colour network's
        # LOGITS: the compositing kernels apply the sigmoid (models/decoder.py:124) on the fly, forward and backward
        Cn, L = self.n_class, self.hid + 1
        check(lib.dns_composite_fwd_ex(ptr(self.raw), ptr(self.z), ptr(self.logit), N, S, Cn, ptr(self.depth), ptr(self.var),
                                       ptr(self.rgb), ptr(self.weights), ptr(self.sem), 1, st), "dns_composite_fwd_ex")
        lam = self.lam
        # (dns_loss_rays -- the rays' sums, the finalize and the rays' backward as ONE single-workgroup launch -- is what the
        #  tracker's 512 rays use; at 4096 rays one workgroup takes 63 us where the three launches take 22: measured, not used here)
        check(lib.dns_loss_sums(lam, N, S, Cn, L, 0, ptr(self.rgb), ptr(self.depth), None, ptr(self.sem), ptr(self.gt_color),
                                ptr(self.gt_depth), ptr(self.gt_label), ptr(self.inside), ptr(fine), ptr(self.coarse),
                                ptr(self.z), ptr(self.sums_ws), st), "dns_loss_sums")
        if self.dist_on:
            m.dist.allreduce_sums(self.sums_ws[:16])
        # (the finalize runs inside the rays' backward kernel below: dns_loss_finalize_bwd)

        # ---- backward.  d_featx [P, 4 + n_feat]: column 3 = d occupancy, columns 4.. = the feature-block gradient of the colour /
        # logit networks, so columns 3 .. 3 + L are the fine network's output gradient in one strided view.  The losses write
        # their d_fine THERE, compositing adds d occupancy, the colour and logit networks add (+=) their feature gradients: no
        # [P, L] sum kernel.  (The 2-D code's columns only ever accumulate; nothing reads them -- the code has no gradient.)
        ldf = 4 + self.n_feat
        d_fine_dst = _V(self.d_featx.data_ptr() + 4 * 3)
        # ray losses -> compositing (its d_raw[:, 0:3] IS the colour network's output gradient: the sigmoid's backward is folded
        # in) -> point losses, which also add the compositing's d occupancy (d_raw[:, 3]) into column 0 of d_fine
        check(lib.dns_loss_finalize_bwd(lam, N, S, Cn, L, 0, ptr(self.sums_ws), ptr(self.out), ptr(self.one), ptr(self.rgb),
                                        ptr(self.depth), None, ptr(self.sem), ptr(self.gt_color), ptr(self.gt_depth),
                                        ptr(self.gt_label), ptr(self.inside), ptr(self.d_color), ptr(self.d_depth), None,
                                        ptr(self.d_sem), st), "dns_loss_finalize_bwd")
        check(lib.dns_composite_bwd_ex(ptr(self.raw), ptr(self.z), ptr(self.logit), N, S, Cn, ptr(self.d_depth), None,
                                       ptr(self.d_color), None, ptr(self.d_sem), ptr(self.d_raw), ptr(self.d_logit), 1, st),
              "dns_composite_bwd_ex")
        check(lib.dns_loss_bwd_points(lam, N, S, Cn, L, ptr(self.out), ptr(self.one), ptr(self.gt_depth), ptr(self.inside), ptr(fine),
                                      ptr(self.coarse), ptr(self.z), d_fine_dst, ptr(self.d_coarse), ldf,
                                      _V(self.d_raw.data_ptr() + 12), 4, st), "dns_loss_bwd_points")

        # dW_in of every network on the SIDE stream (2.02 -> 1.90 ms per step): the streaming kernel is memory-bound and needs only
        # what its backward kernel left in the workspace, the next network's backward kernel is vector-bound -- the pair fills the
        # machine where either alone does not
        fork_dwin = on_side and not half               # (half rows: dW_in is formed inside the ONE backward kernel)
        side_st = _V(self.side.cuda_stream)
        nws = [0]

        def bwd(x2, n_in1, dy, params, shape, d_x2, d_p, ri, tg, n_slots, stride, acc, hs):
            n_in, n_out, nn, nl = shape
            live = 0 if x2 is None else self.live
            ws = self.ws_mlp4[nws[0]] if fork_dwin else self.ws_mlp
            nws[0] += 1
            if half:
                check(lib.dns_mlp_bwd_half(ptr(self.xh), ld, None if x2 is None else ptr(self.feath), self.n_feat, n_in1, ptr(dy),
                                           dy.stride(0), ptr(params), n_in, n_out, nn, nl, ptr(self.d_buf), ld, ptr(d_x2),
                                           0 if d_x2 is None else d_x2.stride(0), ptr(d_p), n_slots, ptr(ri), ptr(tg), stride,
                                           acc | live, self.loss_scale, st), "dns_mlp_bwd_half")
                return
            if sr:
                check(lib.dns_mlp_bwd_split(rows_x, None if x2 is None else rows_f, n_in1, ptr(dy), dy.stride(0), ptr(params), n_in,
                                            n_out, nn, nl, ptr(self.d_buf), ld, ptr(d_x2), 0 if d_x2 is None else d_x2.stride(0),
                                            ptr(d_p), ptr(ws), n_slots, ptr(ri), ptr(tg), stride, acc | fp16, st), "dns_mlp_bwd_split")
                if not fork_dwin:
                    check(lib.dns_mlp_dwin(ptr(self.buf), ld, ptr(x2), 0 if x2 is None else x2.stride(0), n_in1, n_in, nn, nl, ptr(d_p),
                                           ptr(ws), n_slots, ptr(ri), ptr(tg), stride, fp16 | live, st), "dns_mlp_dwin")
            else:
                wp, wflag = self._w(params, n_out)
                check(lib.dns_mlp_bwd(ptr(self.buf), ld, ptr(x2), 0 if x2 is None else x2.stride(0), n_in1, ptr(dy), dy.stride(0),
                                      wp, n_in, n_out, nn, nl, ptr(self.d_buf), ld, ptr(d_x2),
                                      0 if d_x2 is None else d_x2.stride(0), ptr(d_p), ptr(ws), n_slots, ptr(ri), ptr(tg),
                                      stride, ptr(hs), acc | fp16 | wflag | live | (ops.MLP_NO_DWIN_FLAG if fork_dwin else 0), st), "dns_mlp_bwd")
            if fork_dwin:
                # dW_in = dH_1^T x (memory-bound, needs only what this launch left in ws) on the side stream, beside the next
                # network's vector-bound backward kernel
                ev = torch.cuda.Event()
                ev.record(main)
                self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    check(lib.dns_mlp_dwin(ptr(self.buf), ld, ptr(x2), 0 if x2 is None else x2.stride(0), n_in1, n_in, nn, nl, ptr(d_p),
                                           ptr(ws), n_slots, ptr(ri), ptr(tg), stride, fp16 | live, side_st), "dns_mlp_dwin")

        d_feat = self.d_featx[:, 4:]
        bwd(None, 0, self.d_coarse, self.p_coarse, self.shp_c, None, cur.g_coarse, None, None, P, 0, 0, self.h_c)
        bwd(self.feat, pe, self.d_raw, self.p_color, self.shp_col, d_feat, cur.g_color, None, None, P, 0, 3, self.h_col)
        bwd(self.feat, pe, self.d_logit, self.p_logit, self.shp_log, d_feat, cur.g_logit, None, None, P, 0, 3, self.h_log)
        if self.stem:
            # Merge's backward (models/decoder.py:67-77): d code -> the R views' latent gradients -> weight gradients and, under
            # bundle adjustment, d OneBlob -> d(relative point) (added to the points' gradient before the pose reduction)
            n_in_m, n_out_m, nn_m, nl_m = self.shp_m
            pe_m = self.pe_m
            check(lib.dns_merge_dy(_V(self.d_featx.data_ptr() + 4 * (4 + self.hid)), ldf, n_out_m, self.R, self.Pf, ptr(self.z),
                                   ptr(self.gt_depth), N, S, ptr(self.mdy), st), "dns_merge_dy")
            check(lib.dns_mlp_bwd(ptr(self.mbuf), n_in_m, _V(self.mbuf.data_ptr() + 4 * pe_m), n_in_m, pe_m, ptr(self.mdy), n_out_m,
                                  ptr(self.p_merge), n_in_m, n_out_m, nn_m, nl_m, ptr(self.d_mpe), pe_m, None, 0, ptr(cur.g_merge),
                                  ptr(self.ws_m), self.Mr, None, None, 0, None,
                                  ops.MLP_DX_FIRST_FLAG | fp16 | (ops.MLP_NO_DWIN_FLAG if fork_dwin else 0), st), "dns_mlp_bwd")
            if fork_dwin:
                ev = torch.cuda.Event()
                ev.record(main)
                self.side.wait_event(ev)
                with torch.cuda.stream(self.side):
                    check(lib.dns_mlp_dwin(ptr(self.mbuf), n_in_m, None, 0, 0, n_in_m, nn_m, nl_m, ptr(cur.g_merge), ptr(self.ws_m),
                                           self.Mr, None, None, 0, fp16, side_st), "dns_mlp_dwin")
            if self.is_BA:
                check(lib.dns_encode_bwd(ptr(self.xm), self.b6m, self.Mr, self.n_bins_m, None, None, ptr(self.d_mpe), pe_m, None, 0,
                                         None, ptr(self.d_rel), None, None, 0, 0, st), "dns_encode_bwd")
        bwd(None, 0, self.d_featx[:, 3:3 + L], self.p_pool, self.shp_f, None, cur.g_pool, row_index, tile_group,
            self.n_slots, self.p_pool.shape[-1], 1, self.h_f)
        work = None
        if self.dist_on:
            # colour | logit | pool gradients are complete once the LAST forked dW_in has run: with the fork the all-reduce is
            # issued from the side stream (behind that kernel, which itself waited for the fine network's backward kernel)
            import torch.distributed as dist
            if fork_dwin:
                with torch.cuda.stream(self.side):
                    work = dist.all_reduce(cur.G_early, op=dist.ReduceOp.SUM, group=m.dist.group, async_op=True)
            else:
                work = dist.all_reduce(cur.G_early, op=dist.ReduceOp.SUM, group=m.dist.group, async_op=True)
        d_grid = _V(self.d_buf.data_ptr() + 4 * pe)
        fork_pose = self.is_BA and on_side
        if fork_pose:
            # the pose gradient (a streaming kernel over the saved d(features)/dx, then the tiny pose reduction) on the side
            # stream beside the table scatter (vector-bound): 2.036-2.063 -> 2.027-2.047 ms per step
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                st2 = _V(self.side.cuda_stream)
                check(lib.dns_encode_bwd(ptr(self.x3), self.b6, P, self.n_bins, ptr(self.p_table), meta, ptr(self.d_buf), ld, d_grid,
                                         ld, None, ptr(self.d_x3), ptr(self.dydx), None, 0, 0, st2), "dns_encode_bwd")
                if self.stem:
                    check(lib.dns_add_ref_sum(ptr(self.d_rel), self.R, self.Pf, P, ptr(self.d_x3), st2), "dns_add_ref_sum")
                check(lib.dns_raygen_bwd(ptr(pix), ptr(self.Q), self.camv, 0, H, 0, W, K, npf, S, ptr(self.z), ptr(self.d_x3), None,
                                         None, ptr(self.ray_ws), ptr(cur.g_quat), ptr(cur.g_trans), st2), "dns_raygen_bwd")
        check(lib.dns_encode_bwd(ptr(self.x3), self.b6, P, self.n_bins, ptr(self.p_table), meta, ptr(self.d_buf), ld, d_grid, ld,
                                 ptr(cur.g_table), ptr(self.d_x3) if (self.is_BA and not fork_pose) else None, ptr(self.dydx),
                                 ptr(self.ws_enc), self.scatter_form, self.scatter_cap, st), "dns_encode_bwd")
        if self.is_BA and not fork_pose:
            if self.stem:
                check(lib.dns_add_ref_sum(ptr(self.d_rel), self.R, self.Pf, P, ptr(self.d_x3), st), "dns_add_ref_sum")
            check(lib.dns_raygen_bwd(ptr(pix), ptr(self.Q), self.camv, 0, H, 0, W, K, npf, S, ptr(self.z), ptr(self.d_x3), None,
                                     None, ptr(self.ray_ws), ptr(cur.g_quat), ptr(cur.g_trans), st), "dns_raygen_bwd")
        if on_side:
            main.wait_stream(self.side)
        if self.dist_on:
            import torch.distributed as dist
            work2 = dist.all_reduce(cur.G_late, op=dist.ReduceOp.SUM, group=m.dist.group, async_op=True)
            work.wait()
            work2.wait()
        check(lib.dns_adam_step(cur.adam_items, self.n_adam, self.betas[0], self.betas[1], self.eps, ptr(self.adam_state), st),
              "dns_adam_step")
        self.steps += 1

    # ------------------------------------------------------------------------------------------------------------------
    def set_lambda_lt(self, value):
        """Weight of the latent loss for the following steps (slams/mapping.py:893-896: 0 in the first half of an optimize()
        that added decoders)."""
        if getattr(self, "_captured", False) and float(value) != float(self.lam[3]):
            raise RuntimeError("MapStep.set_lambda_lt: the step has been captured into a hipGraph -- the loss weights are kernel "
                               "arguments frozen at capture time; capture one graph per weight schedule phase")
        self.lam[3] = float(value)

    # the last step's gradient segments (tests, inspection)
    g_table = property(lambda self: self.cur.g_table)
    g_coarse = property(lambda self: self.cur.g_coarse)
    g_color = property(lambda self: self.cur.g_color)
    g_logit = property(lambda self: self.cur.g_logit)
    g_pool = property(lambda self: self.cur.g_pool)
    g_merge = property(lambda self: self.cur.g_merge)
    g_quat = property(lambda self: self.cur.g_quat)
    g_trans = property(lambda self: self.cur.g_trans)

    def losses(self):
        """(total, terms) of the last step, as ``Mapper.iteration_loss`` returns them (device tensors, no sync)."""
        t = self.out
        terms = {"p_loss": t[0], "d_loss": t[1], "l_loss": t[2], "lt_loss": t[3], "fs_loss": t[4], "opacity_loss": t[5]}
        total = t[6]
        if self.smooth:
            terms["smooth_loss"] = self.tv[0]
            total = total + self.w_sm[0] * self.tv[0]
        return total, terms

    @torch.no_grad()
    def write_back(self):
        """Poses into the caller's per-frame tensors (``set_optimizer``'s quad_list / T_list)."""
        for k in range(self.K):
            self.quad_list[k].data.copy_(self.Q[k])
            self.T_list[k].data.copy_(self.T[k])

    def pose(self, k):
        bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], device=self.dev)
        return torch.cat([torch.cat((get_rotation_from_quad(self.Q[k]), self.T[k][:, None]), -1), bottom], dim=0)


class TrackStep:
    """The tracker's iteration (slams/tracking.py:313-340: sample -> coarse-only render -> three masked losses -> pose gradient ->
    Adam on (quat, T) -> keep the best pose) as a fixed launch sequence, like ``MapStep``: ~33 launches on one stream over
    buffers allocated once, no host read anywhere (the keep-best comparison runs on the device, ``dns_keep_best``), so ONE
    captured iteration replays n_iters times from a hipGraph.  The scene is frozen (:120-124): no weight-gradient work, no
    table scatter.  Same kernels, arithmetic and generator calls as ``Tracker.track_frame`` (tests/test_gpu_fused_step.py).
    ``features``: None or the per-sample 2-D code [n_pixels, S, C] (stem feature maps run through ``Tracker.track_frame``)."""

    def __init__(self, tracker, cur_frames, est_c2w, features=None, betas=(0.9, 0.999), eps=1e-8, refer_frames=None):
        t = self.t = tracker
        dev = self.dev = torch.device(t.device)
        if dev.type != "cuda":
            raise ValueError("dns_slam_amd ops run on the GPU only; there is no CPU fallback")
        # features: None | per-sample code [N, S, C] | stem maps [1, n_refer, C, h, w] + refer_frames['est_w2c'] [n_refer, 4, 4]:
        # then feature_matching + Decoder.merge run inside every iteration (slams/tracking.py:162-165), Merge frozen like the scene
        self.stem = features is not None and features.dim() == 5
        if features is not None and features.dim() not in (3, 5):
            raise ValueError("TrackStep: features is the per-sample code [N, S, C] or the stem feature maps [1, R, C, h, w]")
        if self.stem and refer_frames is None:
            raise ValueError("TrackStep: stem feature maps need refer_frames['est_w2c']")
        dec = t.decoder
        self.prep = t.prepare_frame(cur_frames)
        self.features = None if (features is None or self.stem) else features.to(dev).float().contiguous()
        self.betas, self.eps = betas, eps
        self.pe_dim, self.grid_dim, self.hid = dec.pe_dim, dec.grid_dim, dec.hidden_dim
        nets = (dec.coarse_fn.decoder, dec.out_fn.color_decoder, dec.out_fn.logit_decoder)
        self.fp16 = ops.MLP_FP16_FLAG if getattr(nets[0], "fp16", False) else 0
        shp = lambda n: (n.n_input_dims, n.n_output_dims, n.n_neurons, n.n_hidden_layers)
        self.shp_c, self.shp_col, self.shp_log = (shp(n) for n in nets)
        self.p_coarse, self.p_color, self.p_logit = (n.params for n in nets)
        self.p_table, self.meta, self.n_bins = dec.pe_fn.grid_fn.params, dec.pe_fn.grid_fn.meta, dec.pe_fn.pe_fn.n_bins
        self.n_feat = self.hid + (self.hid if (features is None or self.stem) else self.features.shape[-1])
        if not (self.pe_dim % 4 == 0 and self.pe_dim <= 64 and self.n_feat <= 64 and self.n_feat % 4 == 0
                and self.shp_col[0] == self.pe_dim + self.n_feat and self.shp_c[1] == self.hid + 1):
            raise ValueError("TrackStep: network shapes outside the two-segment input form")
        self.n_class = self.shp_log[1]
        N = self.N = int(t.n_pixels)
        nu = self.nu = 0 if t.t_uniform is None else t.t_uniform.numel()
        ns = self.ns = t.n_surface_ray
        S = self.S = nu + ns
        P = self.P = N * S
        ld = self.ld = self.pe_dim + self.grid_dim
        f = lambda *s: torch.empty(*s, device=dev, dtype=torch.float32)
        # pose = the parameter; [d_buf | g_quat(4) g_trans(3) pad] share one buffer: one fill clears both per iteration
        self.Q = get_quad_from_c2w(est_c2w).detach().to(dev).float().reshape(1, 4).contiguous()
        self.T = est_c2w[:3, 3].detach().clone().to(dev).float().reshape(1, 3).contiguous()
        self.zbuf = torch.zeros(P * ld + 8, device=dev)
        self.d_buf = self.zbuf[:P * ld].view(P, ld)
        self.g_quat, self.g_trans = self.zbuf[P * ld:P * ld + 4], self.zbuf[P * ld + 4:P * ld + 7]
        self.M, self.V, self.adam_state = torch.zeros(8, device=dev), torch.zeros(8, device=dev), torch.zeros(3, device=dev)
        lr = float(t.cam_lr)
        lrT = lr * 0.2 if t.seperate_LR else lr
        self.adam_items = (DnsAdamTensor * 2)()
        for k, (p, g, o, n, l) in enumerate(((self.T, self.g_trans, 4, 3, lrT), (self.Q, self.g_quat, 0, 4, lr))):
            it = self.adam_items[k]
            it.p, it.g, it.m, it.v, it.n, it.lr = p.data_ptr(), g.data_ptr(), self.M.data_ptr() + 4 * o, self.V.data_ptr() + 4 * o, n, l
        self.best_loss = torch.full((1,), float("inf"), device=dev)
        self.best_cam = torch.cat((self.Q.reshape(-1), self.T.reshape(-1))).clone()
        self.rays_o, self.rays_d, self.gt_color, self.gt_depth = f(N, 3), f(N, 3), f(N, 3), f(N)
        self.gt_label = torch.empty(N, device=dev, dtype=torch.int64)
        self.inside, self.valid = torch.empty(N, device=dev, dtype=torch.uint8), torch.empty(N, device=dev, dtype=torch.uint8)
        self.z, self.pts = f(N, S), f(N, S, 3)
        self.dmax_ws = torch.empty(1, device=dev, dtype=torch.int32)
        self.x3, self.buf = f(P, 3), f(P, ld)
        self.dydx = f(self.meta.n_levels * 3 * P * 2)
        nf = self.hid + 1
        self.lat, self.feat = f(P, nf), f(P, self.n_feat)
        self.raw, self.logit = f(P, 4), f(P, self.n_class)
        self.depth, self.var, self.rgb, self.weights, self.sem = f(N), f(N), f(N, 3), f(N, S), f(N, self.n_class)
        self.sums_ws, self.out, self.one = f(ops.LOSS_SUMS_FLOATS), f(16), torch.ones(1, device=dev)
        self.d_color, self.d_depth, self.d_var, self.d_sem = f(N, 3), f(N), f(N), f(N, self.n_class)
        self.d_raw, self.d_logit, self.d_col = f(P, 4), f(P, self.n_class), f(P, 4)
        self.d_featx = torch.zeros(P, 4 + self.n_feat, device=dev)
        self.d_x3 = f(P, 3)
        raw_lib = ops.lib._raw
        if self.stem:
            _, R, Cs, fh, fw = features.shape
            self.R, self.Cs, self.fh, self.fw = int(R), int(Cs), int(fh), int(fw)
            self.feat_maps = features.detach().to(dev).float()[0].permute(0, 2, 3, 1).contiguous()
            mg = dec.merge.decoder
            self.shp_m, self.p_merge = shp(mg), mg.params
            self.n_bins_m = dec.merge.pe_fn.n_bins
            self.pe_m = 3 * self.n_bins_m
            if not (self.shp_m[0] == self.pe_m + Cs and self.shp_m[1] == self.hid and self.pe_m % 4 == 0):
                raise ValueError("TrackStep: Decoder.merge's network does not match the stem features / hidden width")
            self.b6m = ops._bound6(dec.merge.bound)
            w2c = refer_frames["est_w2c"].clone().detach().to(dev).float()                 # slams/tracking.py:162
            self.w2c = w2c.reshape(R, 16).contiguous()
            self.origin = torch.inverse(w2c)[:, :3, 3].contiguous()                          # utils/common.py:672-674
            Mr = self.Mr = R * P
            self.mbuf, self.rel, self.xm = f(Mr, self.shp_m[0]), f(Mr, 3), f(Mr, 3)
            self.mlat, self.mdy = f(Mr, self.hid), f(Mr, self.hid)
            self.d_mpe, self.d_rel = f(Mr, self.pe_m), f(Mr, 3)
            self.K9 = (C.c_float * 9)(float(t.fx), 0.0, float(t.cx), 0.0, float(t.fy), float(t.cy), 0.0, 0.0, 1.0)
        self.ray_ws = f(max(int(raw_lib.dns_raygen_bwd_ws_floats(1, N)), 1))
        # The scene is frozen for the whole loop: its networks' operand images are built ONCE (DNS_MLP_PREPARED) and every
        # launch copies them in instead of rebuilding them per workgroup (0.253 -> 0.245 ms per iteration).  MapStep does not
        # use this: its weights change every step, and in the two-stream step the five extra launches plus the copy-in traffic
        # cost more than the per-workgroup rebuild (measured on one box: 2.105 without, 2.111 with the images but no
        # re-preparation, 2.133 with both; stand-alone the prepared launches are 1-3 us (forward) and 4-7 us (backward) faster).
        self.PREP = ops.MLP_PREPARED_FLAG
        nets_w = [(self.p_coarse, self.shp_c), (self.p_color, self.shp_col), (self.p_logit, self.shp_log)]
        if self.stem:
            nets_w.append((self.p_merge, self.shp_m))
        self._nets_w = nets_w
        self.w = [f(int(raw_lib.dns_mlp_prepared_floats(*shp_))) for _, shp_ in nets_w]
        self._prepare_weights()
        self.g = None                                      # the captured iteration (run(graph=True)), kept across reset()
        self.w_coarse, self.w_color, self.w_logit = self.w[:3]
        self.w_merge = self.w[3] if self.stem else None
        self.lam = (C.c_float * 8)(t.lambda_p, t.lambda_d, t.lambda_l, 0.0, 0.0, 0.0, 0.0, 1.0)
        self.camv = (C.c_double * 4)(float(t.fx), float(t.fy), float(t.cx), float(t.cy))
        self.b6 = ops._bound6(t.bound)
        self.steps = 0

    @torch.no_grad()
    def step(self, draws=None):
        """One tracking iteration on the current stream (capturable).  ``draws`` = (pix [N] int64, t_surf [ns], t_zero [ns])."""
        t, lib = self.t, ops.lib
        st = _V(torch.cuda.current_stream().cuda_stream)
        N, S, P, ld, pe, b = self.N, self.S, self.P, self.ld, self.pe_dim, t.border
        H, W = t.H, t.W
        if draws is None:                                  # Tracker.draw_pixels / draw_jitter, device generator, same call order
            pix = torch.randint((H - 2 * b) * (W - 2 * b), (N,), device=self.dev)
            t_surf = torch.rand(self.ns, device=self.dev)
            check(lib.dns_force_half(ptr(t_surf), self.ns, self.ns // 2 + 1, st), "dns_force_half")
            t_zero = torch.rand(self.ns, device=self.dev)
        else:
            pix, t_surf, t_zero = draws
        self._keep = (pix, t_surf, t_zero)
        self.zbuf.zero_()
        prep = self.prep
        check(lib.dns_raygen_sample(ptr(pix), ptr(prep["color"]), ptr(prep["depth"]), ptr(prep["label"]), ptr(self.Q), ptr(self.T),
                                    self.camv, self.b6, H, W, b, H - b, b, W - b, 1, N, ptr(t.t_uniform), ptr(t_surf), ptr(t_zero),
                                    self.nu, self.ns, 0, ptr(self.dmax_ws), 0, ptr(self.rays_o), ptr(self.rays_d),
                                    ptr(self.gt_color), ptr(self.gt_depth), ptr(self.gt_label), ptr(self.inside), ptr(self.z),
                                    ptr(self.pts), st), "dns_raygen_sample")
        check(lib.dns_track_mask(ptr(self.gt_depth), ptr(self.inside), N, 0.01, ptr(self.valid), st), "dns_track_mask")   # :171-172
        meta = C.byref(self.meta.c)
        check(lib.dns_encode_fwd(ptr(self.pts), self.b6, P, self.n_bins, ptr(self.p_table), meta, ptr(self.x3), ptr(self.buf), ld,
                                 _V(self.buf.data_ptr() + 4 * pe), ld, ptr(self.dydx), st), "dns_encode_fwd")
        fp16 = self.fp16 | self.PREP

        def fwd(x2, n_in1, params, shape, y):
            n_in, n_out, nn, nl = shape
            check(lib.dns_mlp_fwd(ptr(self.buf), ld, ptr(x2), 0 if x2 is None else x2.stride(0), n_in1, ptr(params), n_in, n_out,
                                  nn, nl, ptr(y), y.stride(0), P, None, None, 0, None, fp16, st), "dns_mlp_fwd")

        fwd(None, 0, self.w_coarse, self.shp_c, self.lat)                  # coarse-only render (slams/tracking.py:196-200)
        if self.stem:
            n_in_m, n_out_m, nn_m, nl_m = self.shp_m
            check(lib.dns_feature_gather_frames(ptr(self.pts), ptr(self.w2c), ptr(self.origin), self.K9, ptr(self.feat_maps), 1, self.R,
                                                P, self.Cs, self.fh, self.fw, H, W, _V(self.mbuf.data_ptr() + 4 * self.pe_m), n_in_m,
                                                ptr(self.rel), st), "dns_feature_gather_frames")
            check(lib.dns_encode_fwd(ptr(self.rel), self.b6m, self.Mr, self.n_bins_m, None, None, ptr(self.xm), ptr(self.mbuf), n_in_m,
                                     None, 0, None, st), "dns_encode_fwd")
            check(lib.dns_mlp_fwd(ptr(self.mbuf), n_in_m, None, 0, 0, ptr(self.w_merge), n_in_m, n_out_m, nn_m, nl_m, ptr(self.mlat),
                                  n_out_m, self.Mr, None, None, 0, None, fp16, st), "dns_mlp_fwd")
            check(lib.dns_feature_block_split(ptr(self.lat), self.hid + 1, self.hid, ptr(self.mlat), self.hid, self.R, P, ptr(self.z),
                                              ptr(self.gt_depth), N, S, ptr(self.feat), self.n_feat, None, 0, None, 0, ptr(self.raw), st),
                  "dns_feature_block_split")
        else:
            check(lib.dns_feature_block(ptr(self.lat), self.hid + 1, self.hid, ptr(self.features), self.n_feat - self.hid, ptr(self.z),
                                        ptr(self.gt_depth), N, S, ptr(self.feat), self.n_feat, ptr(self.raw), st), "dns_feature_block")
        fwd(self.feat, pe, self.w_color, self.shp_col, self.raw)
        fwd(self.feat, pe, self.w_logit, self.shp_log, self.logit)
        check(lib.dns_rgb_sigmoid(ptr(self.raw), P, st), "dns_rgb_sigmoid")
        Cn = self.n_class
        check(lib.dns_composite_fwd(ptr(self.raw), ptr(self.z), ptr(self.logit), N, S, Cn, ptr(self.depth), ptr(self.var),
                                    ptr(self.rgb), ptr(self.weights), ptr(self.sem), st), "dns_composite_fwd")
        lam = self.lam
        # rays' sums + finalize + rays' backward: ONE launch (dns_loss_rays; were four with the 16-word clear)
        check(lib.dns_loss_rays(lam, N, 1, Cn, 1, 1, ptr(self.rgb), ptr(self.depth), ptr(self.var), ptr(self.sem), ptr(self.gt_color),
                                ptr(self.gt_depth), ptr(self.gt_label), ptr(self.valid), None, None, None, ptr(self.sums_ws),
                                ptr(self.out), ptr(self.one), ptr(self.d_color), ptr(self.d_depth), ptr(self.d_var), ptr(self.d_sem),
                                st), "dns_loss_rays")
        check(lib.dns_keep_best(_V(self.out.data_ptr() + 4 * 6), ptr(self.Q), ptr(self.T), ptr(self.best_loss), ptr(self.best_cam),
                                st), "dns_keep_best")                       # :326-335, before the optimiser moves the pose
        # ---- backward to the pose
        check(lib.dns_composite_bwd(ptr(self.raw), ptr(self.z), ptr(self.logit), N, S, Cn, ptr(self.d_depth), ptr(self.d_var),
                                    ptr(self.d_color), None, ptr(self.d_sem), ptr(self.d_raw), ptr(self.d_logit), st),
              "dns_composite_bwd")
        ldf = 4 + self.n_feat
        check(lib.dns_raw_bwd(ptr(self.d_raw), ptr(self.raw), P, ptr(self.d_col), _V(self.d_featx.data_ptr() + 12), ldf, 0, st),
              "dns_raw_bwd")

        def bwd(x2, n_in1, dy, params, shape, d_x2, acc):
            n_in, n_out, nn, nl = shape
            check(lib.dns_mlp_bwd(ptr(self.buf), ld, ptr(x2), 0 if x2 is None else x2.stride(0), n_in1, ptr(dy), dy.stride(0),
                                  ptr(params), n_in, n_out, nn, nl, ptr(self.d_buf), ld, ptr(d_x2),
                                  0 if d_x2 is None else d_x2.stride(0), None, None, P, None, None, 0, None, acc | fp16, st),
                  "dns_mlp_bwd")

        d_feat = self.d_featx[:, 4:]
        bwd(self.feat, pe, self.d_col, self.w_color, self.shp_col, d_feat, 1)        # d_buf (zeroed) += ; feature block =
        bwd(self.feat, pe, self.d_logit, self.w_logit, self.shp_log, d_feat, 3)
        if self.stem:
            n_in_m, n_out_m, nn_m, nl_m = self.shp_m
            pe_m = self.pe_m
            check(lib.dns_merge_dy(_V(self.d_featx.data_ptr() + 4 * (4 + self.hid)), ldf, n_out_m, self.R, P, ptr(self.z),
                                   ptr(self.gt_depth), N, S, ptr(self.mdy), st), "dns_merge_dy")
            check(lib.dns_mlp_bwd(ptr(self.mbuf), n_in_m, _V(self.mbuf.data_ptr() + 4 * pe_m), n_in_m, pe_m, ptr(self.mdy), n_out_m,
                                  ptr(self.w_merge), n_in_m, n_out_m, nn_m, nl_m, ptr(self.d_mpe), pe_m, None, 0, None, None, self.Mr,
                                  None, None, 0, None, ops.MLP_DX_FIRST_FLAG | fp16, st), "dns_mlp_bwd")
            check(lib.dns_encode_bwd(ptr(self.xm), self.b6m, self.Mr, self.n_bins_m, None, None, ptr(self.d_mpe), pe_m, None, 0, None,
                                     ptr(self.d_rel), None, None, 0, 0, st), "dns_encode_bwd")
        bwd(None, 0, self.d_featx[:, 3:3 + self.hid + 1], self.w_coarse, self.shp_c, None, 1)
        check(lib.dns_encode_bwd(ptr(self.x3), self.b6, P, self.n_bins, ptr(self.p_table), meta, ptr(self.d_buf), ld,
                                 _V(self.d_buf.data_ptr() + 4 * pe), ld, None, ptr(self.d_x3), ptr(self.dydx), None, 0, 0, st),
              "dns_encode_bwd")
        if self.stem:
            check(lib.dns_add_ref_sum(ptr(self.d_rel), self.R, P, P, ptr(self.d_x3), st), "dns_add_ref_sum")
        check(lib.dns_raygen_bwd(ptr(pix), ptr(self.Q), self.camv, b, H - b, b, W - b, 1, N, S, ptr(self.z), ptr(self.d_x3), None,
                                 None, ptr(self.ray_ws), ptr(self.g_quat), ptr(self.g_trans), st), "dns_raygen_bwd")
        check(lib.dns_adam_step(self.adam_items, 2, self.betas[0], self.betas[1], self.eps, ptr(self.adam_state), st),
              "dns_adam_step")
        self.steps += 1

    @staticmethod
    def signature(tracker, features, refer_frames):
        """Everything a captured iteration has baked in besides the contents of its buffers: shapes, the constants passed by value
        (loss weights, learning rates, intrinsics, border) and the ADDRESSES of the scene's parameters.  ``Tracker.track_frame``
        keeps one TrackStep per signature and only ``reset``s it from frame to frame."""
        t, dec = tracker, tracker.decoder
        nets = [dec.pe_fn.grid_fn.params, dec.coarse_fn.decoder.params, dec.out_fn.color_decoder.params, dec.out_fn.logit_decoder.params]
        fs = None if features is None else tuple(features.shape)
        if features is not None and features.dim() == 5:
            nets.append(dec.merge.decoder.params)
            fs = fs + (tuple(refer_frames["est_w2c"].shape),)
        #  also baked into a capture: the networks' precision flag, the device, the level table of the grid (passed by
        # value), the ADDRESS of the shared surface jitter, and -- with stem maps -- Merge's bound and OneBlob bin count
        grid = dec.pe_fn.grid_fn
        meta = getattr(grid, "meta", None)
        meta_key = None
        meta = getattr(meta, "c", meta)                      # ops.GridMeta wraps the ctypes DnsGridMeta
        if meta is not None:
            n_lv = int(meta.n_levels)
            meta_key = (n_lv,) + tuple((float(meta.scale[l]), int(meta.resolution[l]), int(meta.size[l]), int(meta.offset[l]),
                                        int(meta.hashed[l])) for l in range(n_lv))
        extra = (bool(getattr(dec.coarse_fn.decoder, "fp16", False)), str(torch.device(t.device)), meta_key,
                 int(dec.pe_fn.pe_fn.n_bins), 0 if t.t_uniform is None else t.t_uniform.data_ptr())
        if features is not None and features.dim() == 5:
            mb = getattr(dec.merge, "bound", None)
            extra = extra + (None if mb is None else tuple(float(v) for v in torch.as_tensor(mb).reshape(-1)),
                             int(getattr(getattr(dec.merge, "pe_fn", None), "n_bins", 0) or 0))
        return extra + (int(t.n_pixels), 0 if t.t_uniform is None else t.t_uniform.numel(), t.n_surface_ray, t.H, t.W, t.border, fs,
                float(t.lambda_p), float(t.lambda_d), float(t.lambda_l), float(t.cam_lr), bool(t.seperate_LR),
                float(t.fx), float(t.fy), float(t.cx), float(t.cy), tuple(float(v) for v in torch.as_tensor(t.bound).reshape(-1)),
                tuple((p.data_ptr(), p.numel()) for p in nets))

    @torch.no_grad()
    def reset(self, cur_frames, est_c2w, features=None, refer_frames=None):
        """The next frame on the SAME buffers (and the same captured graph): new images, initial pose, 2-D code / stem maps and
        reference poses are copied into place, the Adam moments and the keep-best state start over (slams/tracking.py:108-126:
        a new optimiser per frame) and the frozen networks' operand images are rebuilt from the current weights."""
        t, dev = self.t, self.dev
        for k, v in (("color", cur_frames["gt_color"]), ("depth", cur_frames["gt_depth"]), ("label", cur_frames["gt_label"])):
            self.prep[k].copy_(v.to(dev).float()[None])
        if self.stem:
            self.feat_maps.copy_(features.detach().to(dev).float()[0].permute(0, 2, 3, 1))
            w2c = refer_frames["est_w2c"].clone().detach().to(dev).float()
            self.w2c.copy_(w2c.reshape(self.R, 16))
            self.origin.copy_(torch.inverse(w2c)[:, :3, 3])
        elif self.features is not None:
            self.features.copy_(features.to(dev).float())
        self.Q.copy_(get_quad_from_c2w(est_c2w).detach().to(dev).float().reshape(1, 4))
        self.T.copy_(est_c2w[:3, 3].detach().to(dev).float().reshape(1, 3))
        self.M.zero_()
        self.V.zero_()
        self.adam_state.zero_()
        self.best_loss.fill_(float("inf"))
        self.best_cam.copy_(torch.cat((self.Q.reshape(-1), self.T.reshape(-1))))
        self._prepare_weights()
        self.steps = 0

    def _prepare_weights(self):
        st = _V(torch.cuda.current_stream().cuda_stream)
        for (p_, shp_), w in zip(self._nets_w, self.w):
            check(ops.lib.dns_mlp_prepare(ptr(p_), shp_[0], shp_[1], shp_[2], shp_[3], 1, 0, ptr(w), 0, st), "dns_mlp_prepare")
        # the fused kernel's colour / logit images without a 2-D code: the LIVE networks (pe + latent inputs; the zero code columns
        # are not there at all -- no feature block is built, the latent is read out of the coarse network's rows)
        if self.features is None and not self.stem:
            live = ops.MLP_LIVE_IN(self.pe_dim + self.hid)
            if getattr(self, "w_live", None) is None:
                f = lambda n: torch.empty(n, device=self.dev, dtype=torch.float32)
                self.w_live = [f(int(ops.lib._raw.dns_mlp_prepared_floats(self.pe_dim + self.hid, shp_[1], shp_[2], shp_[3])))
                               for shp_ in (self.shp_col, self.shp_log)]
            for (p_, shp_), w in zip(((self.p_color, self.shp_col), (self.p_logit, self.shp_log)), self.w_live):
                check(ops.lib.dns_mlp_prepare(ptr(p_), shp_[0], shp_[1], shp_[2], shp_[3], 1, 0, ptr(w), live, st), "dns_mlp_prepare")

    # ---- the iteration as ONE kernel + a pose kernel (csrc/track_fused.inc, ABI v11) ------------------------------------------
    def fused_supported(self):
        """Can ``run_fused`` take this tracker?  (S <= 64, 64 x 2 or 32 x 1 networks, no in-loop stem branch, fp32-grade MLPs.)"""
        if self.stem or self.fp16 or self.S > 64:
            return False
        if (self.shp_c[2], self.shp_c[3]) not in ((64, 2), (32, 1)):
            return False
        return self._fused_block(1) is not None

    def _fused_block(self, n_iters):
        """The DnsTrackFused argument block for a frame of n_iters iterations (draw arrays allocated here), or None when the
        library refuses the shape."""
        t, dev = self.t, self.dev
        key = int(n_iters)
        fb = getattr(self, "_fb", None)
        if fb is not None and fb["n_iters"] == key:
            return fb
        b, H, W, N, ns = t.border, t.H, t.W, self.N, self.ns
        a = DnsTrackFused()
        keep = {"pix": torch.zeros(key, N, device=dev, dtype=torch.int64), "t_surf": torch.zeros(key, ns, device=dev),
                "t_zero": torch.zeros(key, ns, device=dev), "dmax": torch.zeros(key, device=dev, dtype=torch.int32),
                "iter": torch.zeros(1, device=dev, dtype=torch.int32), "out": torch.zeros(8, device=dev)}
        a.color, a.depth, a.label = self.prep["color"].data_ptr(), self.prep["depth"].data_ptr(), self.prep["label"].data_ptr()
        a.H, a.W, a.H0, a.H1, a.W0, a.W1 = H, W, b, H - b, b, W - b
        a.cam, a.bound = C.addressof(self.camv), C.addressof(self.b6)
        a.pix, a.t_surf, a.t_zero, a.dmax = (keep[k].data_ptr() for k in ("pix", "t_surf", "t_zero", "dmax"))
        a.t_uniform = 0 if t.t_uniform is None else t.t_uniform.data_ptr()
        a.iter, a.n_iters = keep["iter"].data_ptr(), key
        a.n_uniform, a.n_surface, a.n_rays = self.nu, ns, N
        a.quat, a.trans = self.Q.data_ptr(), self.T.data_ptr()
        a.table, a.meta, a.n_bins = self.p_table.data_ptr(), C.pointer(self.meta.c), self.n_bins
        a.w_coarse, a.w_color, a.w_logit = self.w_coarse.data_ptr(), self.w_color.data_ptr(), self.w_logit.data_ptr()
        a.n_neurons, a.n_hidden_layers = self.shp_c[2], self.shp_c[3]
        a.hidden, a.n_feat, a.n_class = self.hid, self.n_feat, self.n_class
        if self.features is not None:
            a.code, a.code_dim = self.features.data_ptr(), self.features.shape[-1]
        else:
            a.code, a.code_dim = 0, 0
            if getattr(self, "w_live", None) is not None and (self.pe_dim + self.hid) % 8 == 0 and self.pe_dim + self.hid > 64:
                # no code: the live networks (no feature block, narrower colour / logit networks)
                a.n_feat = self.hid
                a.w_color, a.w_logit = self.w_live[0].data_ptr(), self.w_live[1].data_ptr()
        a.lambda_p, a.lambda_d, a.lambda_l = float(t.lambda_p), float(t.lambda_d), float(t.lambda_l)
        lr = float(t.cam_lr)
        a.lr_quat, a.lr_trans = lr, (lr * 0.2 if t.seperate_LR else lr)
        a.beta1, a.beta2, a.eps = self.betas[0], self.betas[1], self.eps
        a.adam_m, a.adam_v, a.adam_state = self.M.data_ptr(), self.V.data_ptr(), self.adam_state.data_ptr()
        a.best_loss, a.best_cam = self.best_loss.data_ptr(), self.best_cam.data_ptr()
        a.out, a.g_quat, a.g_trans = keep["out"].data_ptr(), self.g_quat.data_ptr(), self.g_trans.data_ptr()
        n_ws = int(ops.lib._raw.dns_track_fused_ws_floats(C.byref(a)))
        if n_ws == 0:
            return None
        keep["ws"] = torch.empty(n_ws + 64, device=dev)
        off = (-keep["ws"].data_ptr() // 4) % 64                       # 256-byte aligned start
        a.ws = keep["ws"].data_ptr() + 4 * off
        self._fb = {"n_iters": key, "args": a, "keep": keep, "graph": None}
        return self._fb

    @torch.no_grad()
    def run_fused(self, n_iters, graph=False, draws=None):
        """n_iters iterations, two launches each (``dns_track_fused_iter``): the frame's draws are made up front -- ``draws`` =
        (pix [n_iters, N] int64, t_surf [n_iters, ns], t_zero [n_iters, ns]) or None for the device generator (ONE randint and
        two rand calls per frame instead of three launches per iteration) -- and a device counter selects the iteration's rows,
        so every launch of a frame has the same arguments: ``graph`` replays one captured iteration."""
        t, lib = self.t, ops.lib
        fb = self._fused_block(n_iters)
        if fb is None:
            raise ValueError("TrackStep.run_fused: shape outside the fused form (fused_supported() tells)")
        a, keep = fb["args"], fb["keep"]
        st = _V(torch.cuda.current_stream().cuda_stream)
        b, H, W, N = t.border, t.H, t.W, self.N
        if draws is None:
            keep["pix"].copy_(torch.randint((H - 2 * b) * (W - 2 * b), (n_iters, N), device=self.dev))
            keep["t_surf"].copy_(torch.rand(n_iters, self.ns, device=self.dev))
            keep["t_zero"].copy_(torch.rand(n_iters, self.ns, device=self.dev))
        else:
            for k, v in zip(("pix", "t_surf", "t_zero"), draws):
                keep[k].copy_(v.to(self.dev))
        keep["iter"].zero_()
        self._prepare_weights_if_stale()
        check(lib.dns_track_fused_begin(ptr(keep["pix"]), N, n_iters, ptr(self.prep["depth"]), W, b, b, W - b, ptr(keep["t_surf"]), self.ns,
                                        ptr(keep["dmax"]), st), "dns_track_fused_begin")
        if not graph:
            for _ in range(n_iters):
                check(lib._raw.dns_track_fused_iter(C.byref(a), st), "dns_track_fused_iter")
        else:
            if fb["graph"] is None:
                from ._lib import ensure_init
                ensure_init()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    check(lib._raw.dns_track_fused_iter(C.byref(a), _V(torch.cuda.current_stream().cuda_stream)), "dns_track_fused_iter")
                keep["iter"].zero_()                       # (nothing executed during capture)
                fb["graph"] = g
            for _ in range(n_iters):
                fb["graph"].replay()
        self.steps += n_iters
        self.fused_out = keep["out"]
        return self.best_cam, self.best_loss[0]

    def _prepare_weights_if_stale(self):
        pass                                               # (reset() rebuilds the operand images per frame; nothing to do here)

    def run(self, n_iters, graph=True):
        """n_iters iterations -> (best camera tensor [7] = (quat | T), best loss).  ``graph``: capture one iteration into a
        hipGraph and replay it (tracking is 30-50 latency-bound iterations per frame); the capture is kept, so a TrackStep that
        is ``reset`` for the next frame replays without capturing again."""
        if not graph:
            for _ in range(n_iters):
                self.step()
            return self.best_cam, self.best_loss[0]
        if self.g is None:
            from ._lib import ensure_init
            ensure_init()
            self.g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g):                 # records one iteration; nothing executes during capture
                self.step()
        for _ in range(n_iters):
            self.g.replay()
        return self.best_cam, self.best_loss[0]
