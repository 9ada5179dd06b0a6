#!/usr/bin/env python3
"""bench.py -- the mapping optimise iteration of the volumetric-rendering hot path on N MI355X of one node.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full Mapper iteration (reference slams/mapping.py:881-910) on synthetic 640x480 RGB-D+label frames:
draw pixels -> ray generation + depth-guided sampling -> OneBlob + hash grid -> coarse / per-class fine / colour /
logit MLPs -> compositing -> the seven loss terms (incl. the 63^3-point smoothness lattice) -> backward to grid,
MLPs and poses -> (N>1: one flat gradient all-reduce) -> Adam step.  Nothing is skipped inside the timed region.
Workload = BASELINE.json configs[1]: 4096 rays x 64 samples per GPU, T=2^16 hash grid, 2x64 MLPs, mapping only
(--workload cfg3 / cfg5 / cfg5_fp16 / ref: the other configurations).  The iteration is launched eagerly, the smoothness
branch on a second stream.  N GPUs: weak scaling, every rank renders its own 4096 rays (global batch 4096*N);
--union-batch = the N ranks share one 4096-ray batch (strong).

Prints ONE JSON line (rank 0) with `roofline` (the kernel with the largest summed duration: algorithmic bytes or flops /
its duration, from the library's own event pair around every kernel launch on the launch stream -- dns_kernel_timing --
over K eagerly launched steps of the same workload; a replayed graph cannot be bracketed per kernel), `kernel_rooflines`
(the same for every kernel), `iteration_roofline` (whole iteration incl. PMC traffic per step) and `cpu_baseline` (the
oracle, PyTorch CPU, on this host's cores, bounded sample).
"""
import argparse
import math
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    # name: rays/frame (uniform, by-class), n_uniform, n_surface, hash_size, voxel, neurons, layers, smooth_pts
    "cfg2": dict(rays=(683, 341), nu=48, ns=16, hash_size=16, voxel=0.02, nn=64, nl=2, smooth_pts=64,
                 desc="BASELINE configs[1]: room_0 bound, 640x480, 4 target frames, 4096 rays x 64 samples, "
                      "T=2^16 hash grid, 2x64 MLPs, 8 classes + per-class fine decoders, 63^3 smoothness lattice, Adam"),
    "ref": dict(rays=(332, 166), nu=32, ns=15, hash_size=16, voxel=0.02, nn=32, nl=1, smooth_pts=64,
                desc="reference Replica defaults: 1992 rays x 47 samples, 1x32 MLPs"),
    "cfg3": dict(rays=(683, 341), nu=48, ns=16, hash_size=16, voxel=0.02, nn=64, nl=2, smooth_pts=64, bound="office_0", code_seed=5,
                 track_pixels=512,
                 desc="BASELINE configs[2]: office_0 bound, 640x480, semantic head on (8 classes: logit network + per-class fine "
                      "decoders), 2-D feature code U(-1,1) seed 5 on every sample, mapping iteration 4096 rays x 64 samples; the "
                      "tracking loop (512 rays, 50 iterations per frame) is the `tracking` line"),
    "cfg5": dict(rays=(1366, 682), nu=96, ns=32, hash_size=20, voxel=0.04, nn=64, nl=2, smooth_pts=64, bound="scene0000",
                 desc="BASELINE configs[4] shape in fp32: scene0000 bound, 8192 rays x 128 samples, T=2^20 (58.7 MB table), "
                      "2x64 MLPs"),
    "cfg2_fp16": dict(rays=(683, 341), nu=48, ns=16, hash_size=16, voxel=0.02, nn=64, nl=2, smooth_pts=64, mlp_dtype="fp16",
                      desc="configs[1]'s shapes with the MLPs in the REFERENCE's own precision (tcnn: f16, loss scale 128 -- half rows, "
                           "ABI v12); a secondary line: the headline (cfg2) keeps the fp32-grade networks BASELINE's 1e-4 parity asks for"),
    "cfg5_fp16": dict(rays=(1366, 682), nu=96, ns=32, hash_size=20, voxel=0.04, nn=64, nl=2, smooth_pts=64, bound="scene0000",
                      mlp_dtype="fp16",
                      desc="BASELINE configs[4]: scene0000 bound, 8192 rays x 128 samples, T=2^20, 2x64 MLPs in tcnn's own arithmetic "
                           "(HALF ROWS, ABI v12: f16 activations and weight operands on v_mfma_f32_32x32x16_f16, fp32 accumulate, static "
                           "loss scale 128, fp32 master weights and weight gradients); encodings computed in fp32, losses fp32"),
}


F16_MFMA_PEAK_TFLOPS = 2516.6  # MI355X_MICROARCH.md: ~2.5 PF dense f16/bf16 = 1024 flop/clk/SIMD (32x32x16 in 32 clk) x 1024 SIMDs x 2.4 GHz
LDS_ADD_U64_PEAK_GADDS = 64 / 11.9 * 256 * 2.4   # = 3304 G adds/s (see kernel_cost: hashgrid_bwd_pairbins_kernel)
SPLIT_PRODUCTS = 3             # f16 MFMAs per fp32 product on the forward-type products (hi*hi + hi*lo + lo*hi); weight gradients: 6 bf16


def mlp_macs(info):
    n_in, n_out, nn, nl = info["n_in"], info["n_out"], info["nn"], info["nl"]
    return n_in * nn, (nl - 1) * nn * nn, nn * n_out          # first layer, hidden layers, output layer (per point)


def kernel_cost(entry, kernel, units, info, wl):
    """Algorithmic cost of ONE launch of `kernel`: ("hbm", bytes) or ("mfma", fp32-equivalent flops,
    16-bit MFMA flops actually issued for them).  units = points / slots / rays of the launch."""
    S = wl["nu"] + wl["ns"]
    k = kernel.split("<")[0]
    if k in ("encode_fwd_kernel", "encode_fwd_split_kernel"):
        return ("hbm", units * (16 * 8 * 2 * 4 + 12))                     # 128 table cells of 8 B gathered + the point
    if k == "encode_bwd_kernel":
        return ("hbm", units * (16 * 3 * 8 + 80 * 4 + 12 + 12))            # the forward's dy_dx + the gradient row + x in, d_x out
    if k == "dgrid_transpose_kernel":
        return ("hbm", units * 2 * 32 * 4)                                 # [P,32] gradient read, level-major copy written
    # table scatter, per point and level: RMW of 8 cells of 8 B + the level's 8 B of gradient (+ the point, 12 B, per kernel).  Pair
    # lists (the hashed levels): pass 1 reads point + gradient twice and writes four 4-byte entries, pass 2 reads them back
    L = (info or {}).get("n_levels", 16)
    n_list = (info or {}).get("list_levels", 0)
    if k in ("hashgrid_bwd_binned_kernel", "hashgrid_bwd_queue_kernel"):
        return ("hbm", units * ((L - n_list) * (2 * 8 * 8 + 8) + 12))
    if k == "hashgrid_bwd_pairbins_kernel":
        # The read-modify-write of the table rows happens in LDS bins (64-bit fixed-point ds_add_u64), not in memory: pricing it
        # as HBM traffic (round 4) gave "6.4 TB/s" against 89 MB of PMC bytes.  Yardstick = the LDS atomic rate measured with
        # tools/lds_atomic_rate.hip: one ds_add_u64 wave-instruction per 11.9 cycles per CU; an entry (one x-pair of a point-level:
        # 4 entries per point-level) adds 2 rows x 2 features = 4 values.
        return ("lds_atomic", units * n_list * 4 * 4)
    if k == "hashgrid_bwd_pairlist_kernel":
        return ("hbm", units * n_list * (2 * (12 + 8) + 16))
    if k in ("composite_fwd_kernel", "composite_bwd_kernel"):
        return ("hbm", units * S * (4 + 1 + 8) * 4 * (2 if k == "composite_bwd_kernel" else 1))   # raw + z + logits (and their gradients)
    if k == "mlp_fwd_kernel":
        m = sum(mlp_macs(info))
        return ("mfma", 2 * units * m, 2 * units * m * (1 if wl.get("mlp_dtype") == "fp16" else SPLIT_PRODUCTS))
    if k == "mlp_bwd_kernel":
        m_in, m_hid, m_out = mlp_macs(info)
        single = wl.get("mlp_dtype") == "fp16"
        pf, pw = (1, 3) if single else (SPLIT_PRODUCTS, 6)
        # ALGORITHMIC backward work only: the dH chain, dX and the weight gradients this kernel produces.  The recomputation of
        # the hidden activations (m_in + m_hid: an implementation choice that saves their HBM round trip) is work the kernel
        # ISSUES, not work the backward pass requires -- it counts in mfma_issue_frac, never in `achieved` / `frac`.
        recompute = m_in + m_hid
        chain = (m_out + m_hid) + (m_in if info["dx"] else 0)                       # dH chain, dX
        wgrad = (m_out + m_hid) if info["dw"] else 0                                # dW_out, dW_hidden (dW_in: mlp_dwin_kernel)
        return ("mfma", 2 * units * (chain + wgrad), 2 * units * ((recompute + chain) * pf + wgrad * pw))
    # half rows (ABI v12, BASELINE configs[4]): f16 rows in, fp32 rows out, ONE backward kernel per network -- streaming kernels by
    # design (64 kFLOP per point against ~600 B): their bound is HBM.  Algorithmic bytes: the f16 input row, the fp32 output /
    # output-gradient row, the fp32 input-gradient row written (a read-add-write launch's read is the implementation's)
    if k == "mlp_half_fwd_kernel":
        return ("hbm", units * (2 * info["n_in"] + 4 * info["n_out"]))
    if k == "mlp_half_bwd_kernel":
        n_dx = max(info["n_in"] - info.get("dx_from", 0), 0) if info["dx"] else 0
        return ("hbm", units * (2 * info["n_in"] + 4 * info["n_out"] + 4 * n_dx))
    if k == "mlp_dwin_kernel":
        # a STREAMING kernel (x and dH_1 read once, 96-128 accumulator registers, no reuse): its honest bound is HBM
        return ("hbm", units * (info["n_in"] + info["nn"]) * 4)
    # the streaming kernels of the losses and of the iteration's glue (units = ray-samples of the launch)
    if k == "loss_point_sums_kernel":
        return ("hbm", units * 2 * 33 * 4)                                 # fine + coarse latents read
    if k == "loss_point_bwd_kernel":
        return ("hbm", units * 4 * 33 * 4)                                 # ... read again, their gradients written
    if k in ("feature_block_kernel", "feature_block_split_kernel"):
        # fine latent row read; with a 2-D code: the code read and [latent | masked code] = 64 columns written; without one the
        # code columns are identically zero and neither written nor read (DNS_MLP_LIVE_IN): 32 columns written
        coded = wl.get("code_seed") is not None
        return ("hbm", units * (33 * 4 + (32 * 4 if coded else 0) + (64 if coded else 32) * 4))
    if k == "raw_bwd_kernel":
        return ("hbm", units * (3 * 16 + 8))
    if k == "rgb_sigmoid_kernel":
        return ("hbm", units * 2 * 16)
    return None


def kernel_rooflines(spans, wl, steps, pmc):
    """Per-kernel roofline rows from the library's own event spans (one per launch, on the launch stream)."""
    agg = {}
    for entry, kernel, ms, units, info in spans:
        c = kernel_cost(entry, kernel, units, info, wl)
        row = agg.setdefault(kernel.split("<")[0], {"launches": 0, "ms": 0.0, "bytes": 0, "flops": 0, "issued": 0, "modelled": c is not None})
        row["launches"] += 1
        row["ms"] += ms
        if c and c[0] == "hbm":
            row["bytes"] += c[1]
        elif c and c[0] == "lds_atomic":
            row["lds_adds"] = row.get("lds_adds", 0) + c[1]
        elif c:
            row["flops"] += c[1]
            row["issued"] += c[2]
    rows = []
    for k, r in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        out = {"kernel": k, "launches_per_step": r["launches"] / steps, "ms_per_step": r["ms"] / steps, "avg_launch_ms": r["ms"] / r["launches"]}
        sec = r["ms"] / 1e3
        if r["bytes"]:
            ach = r["bytes"] / sec / 1e9
            out.update({"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_launch": r["bytes"] / r["launches"]})
        elif r.get("lds_adds"):
            ach = r["lds_adds"] / sec / 1e9
            out.update({"bound": "lds_atomic", "achieved": ach, "peak": LDS_ADD_U64_PEAK_GADDS, "unit": "G adds/s",
                        "frac": ach / LDS_ADD_U64_PEAK_GADDS, "algorithmic_adds_per_launch": r["lds_adds"] / r["launches"],
                        "peak_note": "64-bit fixed-point LDS atomics (ds_add_u64): 64 adds per wave-instruction, one per 11.9 cycles per "
                                     "CU (tools/lds_atomic_rate.hip) x 256 CUs x 2.4 GHz; HBM bytes of the kernel: `traffic`"})
        elif r["flops"]:
            ach = r["flops"] / sec / 1e12
            peak = F16_MFMA_PEAK_TFLOPS / (1 if wl.get("mlp_dtype") == "fp16" else SPLIT_PRODUCTS)
            out.update({"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                        "algorithmic_flops_per_launch": r["flops"] / r["launches"],
                        "mfma_issue_frac": r["issued"] / sec / 1e12 / F16_MFMA_PEAK_TFLOPS,
                        "peak_note": "fp32-equivalent flops; peak = 16-bit dense MFMA peak / products per fp32 product "
                                     "(3 x f16 forward-type, 6 x bf16 weight gradients: mfma_issue_frac counts those)"})
        ent = (pmc or {}).get("kernels", {}).get(k)
        out["traffic"] = (ent["fetch_bytes"] + ent["write_bytes"]) if ent else None
        rows.append(out)
    return rows


def build(wl, device, seed, dist_ctx, overlap=False, graph=False, prefetch=None, fused_step=True, stem_features=False):
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.mapping import Mapper
    scene_bound = {"scene0000": synthetic.SCENE0000_BOUND, "office_0": synthetic.OFFICE0_BOUND}.get(wl.get("bound"))
    bound, cam, frames = synthetic.make_scene(4, seed=0, bound=scene_bound)
    n_per_frame = sum(wl["rays"])
    cfg = synthetic.default_cfg(n_pixels=4 * n_per_frame, n_samples_ray=wl["nu"], n_surface_ray=wl["ns"], n_frames=4,
                                hash_size=wl["hash_size"], voxel_size=wl["voxel"], n_neurons=wl["nn"],
                                n_hidden_layers=wl["nl"], smooth_pts=wl["smooth_pts"], mlp_dtype=wl.get("mlp_dtype", "fp32"),
                                track_pixels=wl.get("track_pixels", 500))
    torch.manual_seed(1234)                                  # identical initial parameters on every rank
    dec = Decoder(cfg["model"], bound, n_class=8).to(device)
    mapper = Mapper(cfg, dec, bound, cam, device=device)
    mapper.rays_per_frame = wl["rays"]
    mapper.dist = dist_ctx
    mapper.is_BA = True
    mapper.static_shapes = True                              # sync-free iteration: capturable in a hipGraph
    mapper.overlap_smooth = overlap
    # the next iteration's pixel / jitter / lattice draws are enqueued on the side stream (same generator order)
    # (a captured iteration makes its own draws: nothing to run ahead; `prefetch` overrides, for the A/B test)
    mapper.prefetch_draws = (overlap and not graph) if prefetch is None else bool(prefetch)
    mapper.set_decoder(frames)
    optimizer, quad_list, T_list = mapper.set_optimizer(frames, fused=True)     # csrc/adam.hip: one launch, step count on device
    for grp, lr in zip(optimizer.param_groups, (mapper.lr, mapper.BA_cam_lr, mapper.BA_cam_lr)):
        grp["lr"] = lr
    prep = mapper.prepare_frames(frames)
    torch.manual_seed(seed)                                  # per-rank ray draws
    torch.cuda.manual_seed(seed)
    params = [p for g in optimizer.param_groups for p in g["params"]]
    buckets = None
    if dist_ctx.enabled and not graph:
        # persistent gradient buckets, all-reduced asynchronously from autograd hooks (dist.py): the colour / logit /
        # fine-decoder gradients are complete when the ray branch's MLP backward ends and travel under the hash-grid
        # scatter; the table, the coarse network (both also fed by the lattice branch) and the poses go last
        early = [dec.out_fn.color_decoder.params, dec.out_fn.logit_decoder.params, mapper.fine_decoders.pool]
        buckets = dist_ctx.make_buckets([early, [p for p in params if all(p is not e for e in early)]], hook_launch=[True, False])
    code = None
    if wl.get("code_seed") is not None:      # 2-D feature code of every sample (U(-1,1), seed 5), resident in HBM
        g = torch.Generator().manual_seed(wl["code_seed"])
        code = (torch.rand(4 * n_per_frame, wl["nu"] + wl["ns"], 32, generator=g) * 2 - 1).to(device)

    mapper.bench_code = code
    refer = None
    if stem_features:
        # the reference's REAL iteration: feature_matching + Decoder.merge inside every step (slams/mapping.py:532-557) on the
        # frozen image stem's maps of three reference views per target frame (two keyframes + the frame itself, :399-419);
        # stem = conv 7x7 / 2 + BN + ReLU of models/encoder.py:9-17 with random-init weights (no checkpoint offline), run ONCE
        from dns_slam_amd.encoder import ResNet
        frames = dict(frames)
        frames["kf_idx"] = [0, 10, 20, 30]
        refs = [[98, 99, -1], [0, 99, -1], [10, 0, -1], [20, 10, -1]]            # foreign keyframes keep their stored pose
        src_of = lambda i, rid: i if rid == -1 else (frames["kf_idx"].index(rid) if rid in frames["kf_idx"] else (i + 1) % 4)
        refer = {"kf_idx": refs,
                 "gt_color": torch.stack([torch.stack([frames["gt_color"][src_of(i, r)] for r in refs[i]]) for i in range(4)]),
                 "est_c2w": torch.stack([torch.stack([frames["est_c2w"][src_of(i, r)] for r in refs[i]]) for i in range(4)])}
        stem = ResNet(seed=3).to(device)
        code = stem(refer["gt_color"].to(device)).detach()                      # [4, 3, 64, H/2, W/2]
    if fused_step:
        # the iteration as a fixed launch sequence over preallocated buffers (dns_slam_amd/fused_step.py): same kernels and
        # draws as the autograd-driven step below, without autograd's glue launches and host time
        from dns_slam_amd.fused_step import MapStep
        ms = mapper.map_step = MapStep(mapper, frames, quad_list, T_list, prep=prep, features=code, lambda_lt=10.0, smooth=True,
                                       refer_frames=refer)

        def step():
            ms.step()
            return ms.out[6]                 # the ray-batch loss (a view: no launch); ms.losses() adds the smoothness term

        return cfg, bound, cam, frames, mapper, step

    def step():
        if buckets is not None:
            buckets.zero()
        else:
            optimizer.zero_grad(set_to_none=True)
        samples = mapper.get_target_samples(frames, quad_list, T_list, prep=prep, features=code, refer_frames=refer)
        loss, terms = mapper.iteration_loss(samples, lambda_lt=10.0, smooth=True)
        loss.backward()
        if buckets is not None:
            buckets.finish()
        else:
            dist_ctx.allreduce_grads(params)
        optimizer.step()
        return loss

    return cfg, bound, cam, frames, mapper, step


def capture(step, n_warm=3):
    """Capture one full iteration (sampling -> ... -> Adam) into a hipGraph; replays draw fresh rays (graph-safe Philox
    offsets).  Returns a zero-argument callable returning the (static) loss tensor.  Valid since the library stopped issuing
    hipMemsetAsync (memset nodes of a captured graph stop clearing their destination after a host synchronisation on ROCm
    7.0.51831 -- DESIGN.md section 5), but not the default: a graph's branches do not run concurrently on this ROCm (round 3: 2.28 ms
    replayed against 1.90 launched eagerly on two streams)."""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(n_warm):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = step()

    def replay():
        g.replay()
        return loss                  # the graph's static output: holds the last replay's loss
    return replay


def slam_loop(cfg, bound, cam, frames, mapper, step, device, n_frames, map_every=5, map_iters=100):
    """BASELINE configs[2] as a LOOP (reference slams/dns_slam.py:161-172 runs Tracker and Mapper as two processes on one GPU):
    every frame is tracked (n_iters pose iterations against a frozen copy of the scene, slams/tracking.py:81,313-340), every
    `map_every`-th frame starts `map_iters` mapping iterations (slams/mapping.py:881-910).  One process, two streams: the
    tracker's launches go to a high-priority stream, the mapper's to a second one, and the host interleaves their enqueues
    (map_iters / map_every mapping iterations behind each tracked frame), so both streams always have work queued -- the two step
    drivers of this package in a for-loop, no control plane.  After a keyframe's last mapping iteration the tracker's copy of
    the scene is refreshed in place (the reference's update_para_from_mapping)."""
    import copy
    from dns_slam_amd.tracking import Tracker
    trk_dec = copy.deepcopy(mapper.decoder)
    tracker = Tracker(dict(cfg), trk_dec, bound, cam, device=device)
    tracker.use_track_step = True
    n_it = cfg["tracking"]["n_iters"]
    s_trk, s_map = torch.cuda.Stream(priority=-1), torch.cuda.Stream()
    src = [p for p in mapper.decoder.parameters()]
    dst = [p for p in trk_dec.parameters()]
    per_frame = -(-map_iters // map_every)

    def run(nf, record):
        pending = 0
        ev = []
        for f in range(nf):
            k = f % 4
            cur = {"gt_color": frames["gt_color"][k], "gt_depth": frames["gt_depth"][k], "gt_label": frames["gt_label"][k]}
            with torch.cuda.stream(s_trk):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                tracker.track_frame(cur, frames["est_c2w"][k], n_iters=n_it, graph=False)
                e1.record()
                if record:
                    ev.append(("trk", e0, e1))
            if f % map_every == 0:
                pending = map_iters
                m0 = torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(s_map):
                    m0.record()
            if pending:
                with torch.cuda.stream(s_map):
                    n = min(pending, per_frame)
                    for _ in range(n):
                        step()
                    pending -= n
                    if pending == 0:
                        with torch.no_grad():
                            torch._foreach_copy_(dst, src)            # the tracker's scene copy, refreshed in place
                        m1 = torch.cuda.Event(enable_timing=True)
                        m1.record()
                        if record:
                            ev.append(("map", m0, m1))
                        s_trk.wait_stream(s_map)                      # the next tracked frame sees the refreshed copy
        return ev

    cur_stream = torch.cuda.current_stream()
    s_trk.wait_stream(cur_stream)
    s_map.wait_stream(cur_stream)
    run(map_every, False)                                             # warm-up: one keyframe period
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev = run(n_frames, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    trk = sorted(a.elapsed_time(b) for kind, a, b in ev if kind == "trk")
    mp = sorted(a.elapsed_time(b) for kind, a, b in ev if kind == "map")
    med = lambda v: v[len(v) // 2] if v else None
    return {"frames": n_frames, "frames_per_s": n_frames / dt, "ms_per_frame_wall": dt * 1e3 / n_frames,
            "tracker_ms_per_frame": med(trk), "tracker_iters_per_frame": n_it, "tracker_rays": cfg["tracking"]["n_pixels"],
            "mapper_ms_per_keyframe": med(mp), "mapper_iters_per_keyframe": map_iters, "keyframe_every": map_every,
            "what": "tracker on a high-priority stream, mapper (this line's mapping iteration) on a second stream of the same process; "
                    "stream spans are event pairs on each stream while BOTH run (they stretch each other); frames_per_s is host wall time"}


def host_cores():
    """CPU threads this process may actually use: min(affinity, cgroup quota) -- os.cpu_count() reports the whole node."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return n


def cpu_baseline(wl, cfg, bound, cam, frames, budget_s=25.0):
    """The oracle (PyTorch CPU fp32 restatement, oracle/) timed on this host: full mapping iterations of the SAME
    workload (sample -> render -> 7 losses -> backward), as many as fit the budget (>= 1)."""
    from oracle import slam_ref as sr
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import oracle_cfg_from
    cores = min(host_cores(), 32)
    torch.set_num_threads(cores)
    om = sr.OracleModel(oracle_cfg_from(cfg, 8), bound, fine_classes=frames["label_dict"])
    camt = (cam["H"], cam["W"], cam["fx"], cam["fy"], cam["cx"], cam["cy"])
    from dns_slam_amd.common import get_quad_from_c2w
    quats = [get_quad_from_c2w(frames["est_c2w"][f]) for f in range(4)]
    Ts = [frames["est_c2w"][f][:3, 3].clone() for f in range(4)]
    img5 = [torch.cat((frames["gt_color"][f], frames["gt_depth"][f][..., None], frames["gt_label"][f][..., None]), -1)
            for f in range(4)]
    npf = sum(wl["rays"])
    lc = sr.LossCfg(smooth_pts=wl["smooth_pts"])
    g = torch.Generator().manual_seed(0)
    code = None
    if wl.get("code_seed") is not None:
        code = torch.rand(4 * npf, wl["nu"] + wl["ns"], 32, generator=torch.Generator().manual_seed(wl["code_seed"])) * 2 - 1
    times = []
    t_start = time.perf_counter()
    warm = True                                  # one untimed warm-up iteration (allocator, thread pool), then timed ones
    while True:
        t0 = time.perf_counter()
        om.zero_grad()
        fr = []
        for f in range(4):
            idx = torch.randint(cam["H"] * cam["W"], (npf,), generator=g)
            t = torch.rand(wl["ns"], generator=g)
            t[wl["ns"] // 2 + 1] = 0.5
            fr.append(sr.frame_samples(img5[f], quats[f], Ts[f], camt, bound, idx, t, torch.rand(wl["ns"], generator=g),
                                       wl["nu"], wl["ns"], features=None if code is None else code[f * npf:(f + 1) * npf]))
        so = sr.mapper_target_samples(fr)
        loss, _, _ = sr.mapping_loss(om, so, lc, torch.rand(3, generator=g), torch.rand((1, 1, 1, 3), generator=g))
        loss.backward()
        dt = time.perf_counter() - t0
        if warm:
            warm = False
            if time.perf_counter() - t_start + 2 * dt <= budget_s:
                continue                         # (a host too slow for two iterations inside the budget keeps its only one)
        times.append(dt)
        if time.perf_counter() - t_start + dt > budget_s or len(times) >= 20:
            break
    ts = sorted(times)
    med = ts[len(ts) // 2] if len(ts) % 2 else 0.5 * (ts[len(ts) // 2 - 1] + ts[len(ts) // 2])
    n_samples = 4 * npf * (wl["nu"] + wl["ns"])
    return {"value": n_samples / med, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "sample": f"full mapping iterations of the same workload ({4 * npf} rays x {wl['nu'] + wl['ns']} samples + "
                      f"{wl['smooth_pts'] - 1}^3 smoothness lattice, fwd+bwd, no optimiser step): 1 warm-up + {len(times)} timed, "
                      f"MEDIAN {med * 1e3:.0f} ms/iter (min {ts[0] * 1e3:.0f}); bounded to ~{budget_s:.0f} s of CPU work, so fewer "
                      f"than SURVEY 8d's 5 warm-up + 20 timed iterations; torch {torch.__version__} CPU fp32, {cores} threads"}


def relay_rank0_line(cmd, n_gpus, env=None):
    """Parent side of ``bench.py --gpus N`` (N > 1, not yet under a launcher): run ``cmd`` (the one-process-per-GPU launcher) as a
    CHILD process, pass its stderr through, and relay rank 0's JSON line -- but only if the child exited 0 and the line says the
    job really ran on ``n_gpus`` ranks.  Returns the exit code for the parent.  The parent never touches the GPU (it must not:
    its children own the devices)."""
    import subprocess
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    text = p.stdout.decode(errors="replace")
    line = None
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                cand = json.loads(ln)
            except ValueError:
                continue
            if isinstance(cand, dict) and "metric" in cand:
                line = (ln, cand)
    if p.returncode != 0:
        sys.stderr.write(text)
        print(f"[bench] the {n_gpus}-rank launch exited with code {p.returncode}: no result line", file=sys.stderr, flush=True)
        return p.returncode
    if line is None:
        sys.stderr.write(text)
        print(f"[bench] the {n_gpus}-rank launch printed no result line", file=sys.stderr, flush=True)
        return 3
    if line[1].get("n_gpus") != n_gpus or line[1].get("rccl_ranks", line[1].get("n_gpus")) != n_gpus:
        print(f"[bench] asked for {n_gpus} ranks, the launch reports n_gpus={line[1].get('n_gpus')} "
              f"rccl_ranks={line[1].get('rccl_ranks')}: refusing to relay a line for a different job size", file=sys.stderr, flush=True)
        return 4
    print(line[0], flush=True)
    return 0


def launch_ranks(n_gpus, argv):
    """``python bench.py --gpus N`` outside a launcher: start ``python -m torch.distributed.run --nproc-per-node N bench.py ...`` as a
    child (one rank per GPU over RCCL; rendezvous on 127.0.0.1) and relay rank 0's line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return relay_rank0_line(cmd, n_gpus, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--verbose", action="store_true", help="per-step progress on stderr")
    ap.add_argument("--no-render-forward", action="store_true", help="skip the secondary full-image render line")
    ap.add_argument("--graph", action="store_true",
                    help="replay the iteration from a hipGraph on one stream (slower than the eager two-stream default, see capture())")
    ap.add_argument("--eager", "--no-graph", dest="eager", action="store_true",
                    help="(default) launch the iteration eagerly, the smoothness branch on a second stream")
    ap.add_argument("--no-overlap", action="store_true", help="eager, but keep the smoothness branch on the main stream")
    ap.add_argument("--graph-branches", action="store_true", help="with --graph: capture the smoothness branch as a parallel branch")
    ap.add_argument("--autograd-step", action="store_true",
                    help="drive the iteration through torch.autograd (Mapper.iteration_loss + backward + FusedAdam) instead of "
                         "the fixed launch sequence of dns_slam_amd/fused_step.py")
    ap.add_argument("--stem-features", action="store_true",
                    help="run the 2-D feature branch INSIDE every iteration (feature_matching + Decoder.merge on stem feature maps of 3 "
                         "reference views per target frame, slams/mapping.py:532-557) instead of a precomputed per-sample code")
    ap.add_argument("--loop", type=int, default=0, metavar="F",
                    help="after the timed region: the interleaved SLAM loop of BASELINE configs[2] over F frames (track every frame, "
                         "map every 5th; reported as `slam_loop`)")
    ap.add_argument("--union-batch", action="store_true",
                    help="N>1: the N ranks share ONE batch of the configured size (shared-seed draws, rank slices of the rays and "
                         "of the smoothness lattice; strong scaling) instead of one batch per rank (weak scaling, the default)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher yet: this process becomes the parent of N ranks and touches no GPU itself
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    from dns_slam_amd import dist as ddist
    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # intended: the lattice branch's stream
    from dns_slam_amd import ops
    ctx = ddist.init_from_env(mode="union" if args.union_batch else "weak")
    if ctx.world_size != args.gpus:
        # a line for a job of another size than the one asked for would be read as the asked-for size: refuse
        print(f"[bench] --gpus {args.gpus} but the process group has {ctx.world_size} rank(s) (WORLD_SIZE={os.environ.get('WORLD_SIZE')}): "
              "launch one rank per GPU (python bench.py --gpus N does it itself)", file=sys.stderr, flush=True)
        sys.exit(5)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DNS_FORCE_DEVICE") is not None:       # rehearsal of the N-rank path on a one-GPU box (with gloo)
        local = int(os.environ["DNS_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    wl = WORKLOADS[args.workload]
    # Launch mode.  Eager launches on two streams overlap the smoothness branch with the ray branch; a hipGraph replay needs no
    # host work but runs the branches one after the other (graph branches do not run concurrently on this ROCm: 2.46 with the
    # lattice captured as a branch, 2.45 without).  With the fixed launch sequence (fused_step.MapStep: 0.44 ms of host enqueue
    # per step) eager wins on every workload measured (cfg2 2.09 vs 2.28-2.34 ms, ref 1.00 vs 1.23); with the autograd driver
    # (--autograd-step: 2.0 ms of enqueue) small workloads on a slow host were HOST-bound and the graph won.  Default (neither
    # --graph nor --eager, one GPU): both are built, timed for a few untimed steps, and the faster one runs the timed region
    # -- reported as launch_mode / launch_trial_ms.
    auto_mode = not args.graph and not args.eager and not args.no_overlap and ctx.world_size == 1
    use_graph = args.graph and not args.eager and ctx.world_size == 1
    # the smoothness branch on a second stream (eager), or -- with --graph-branches -- as a parallel branch of the captured graph
    overlap = (not use_graph or args.graph_branches) and not args.no_overlap
    union = ctx.union
    cfg, bound, cam, frames, mapper, step = build(wl, device, seed=100 + (0 if union else ctx.rank), dist_ctx=ctx, overlap=overlap,
                                                  graph=use_graph, fused_step=not args.autograd_step, stem_features=args.stem_features)
    n_rays = 4 * sum(wl["rays"])
    job_rays = n_rays if union else n_rays * ctx.world_size          # rays the whole job renders per step
    S = wl["nu"] + wl["ns"]

    # The step's chain of dependent kernels runs on a HIGH-priority stream, the side stream (lattice branch, next step's
    # preparation) keeps the default priority: where both have workgroups ready the critical path goes first (the device has two
    # levels; 2.063-2.067 -> 2.048-2.059 ms per step, tools/stream_priority.py; the other way round: 2.115)
    hp = torch.cuda.Stream(priority=-1)
    hp.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(hp)
    run = step
    graphed = False
    if use_graph:
        try:
            run = capture(step)
            graphed = True
        except Exception as e:                               # keep the benchmark alive: fall back to eager launches
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr, flush=True)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            run = step
    trial = None
    if auto_mode:
        def timed(fn, n=8):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) * 1e3 / n
        trial = {"eager_2_streams": timed(step)}
        try:
            mapper.overlap_smooth, mapper.prefetch_draws = False, False      # a captured iteration makes its own draws, one stream
            mapper._pending_draws = None
            replay = capture(step)
            trial["graph"] = timed(replay)
            if trial["graph"] < trial["eager_2_streams"]:
                run, graphed, overlap = replay, True, False
        except Exception as e:
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eagerly", file=sys.stderr, flush=True)
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
        if not graphed:
            mapper.overlap_smooth, mapper.prefetch_draws = True, True
            mapper._pending_draws = None
            replay = None                                   # the losing graph and its memory pool go away before the timed region
            import gc
            gc.collect()
            torch.cuda.synchronize()
            for _ in range(10):
                step()
            torch.cuda.synchronize()
    for i in range(args.warmup):
        tw = time.perf_counter()
        run()
        if args.verbose:
            torch.cuda.synchronize()
            print(f"[bench] warmup step {i}: {(time.perf_counter() - tw) * 1e3:.1f} ms", file=sys.stderr, flush=True)
    # The interpreter's cyclic garbage collector is parked for the timed region (as timeit does): a generation-2 pass over the
    # autograd objects of a step showed up as one 55-60 ms step in a hundred (step_ms_spread.max), i.e. +0.6 ms on the mean.
    import gc
    gc.collect()
    gc.disable()
    ctx.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    marks = []
    for i in range(args.steps):
        last = run()
        if args.verbose and i % 10 == 9:
            torch.cuda.synchronize()
            marks.append(time.perf_counter() - t0)
    if marks:
        print("[bench] cumulative s at every 10th timed step:", [round(m, 4) for m in marks], file=sys.stderr, flush=True)
    ctx.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if getattr(mapper, "map_step", None) is not None:
        last = mapper.map_step.losses()[0]
    final_loss = float(last.detach()) if torch.is_tensor(last) else None        # after the clock stopped
    if final_loss is not None and not math.isfinite(final_loss):
        raise RuntimeError(f"non-finite loss after the timed steps ({final_loss}): the measurement is void")
    elapsed = ctx.max_over_ranks(elapsed, device)
    # spread: the same K steps once more with an event pair around every step (outside the timed region; the events sit on the
    # main stream, which every step's side-stream work joins before the optimiser step)
    spread = None
    if ctx.world_size == 1:
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for e0, e1 in evs:
            e0.record()
            run()
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        spread = {"min": ts[0], "median": ts[len(ts) // 2], "max": ts[-1], "p90": ts[int(0.9 * (len(ts) - 1))],
                  "note": "per-step event spans of a second pass of the same K steps (enqueue-ahead makes a single step's span "
                          "shorter than ms_per_step when the host runs ahead of the GPU)"}
    # per-kernel durations for the roofline: the same K steps launched eagerly with an event pair around every
    # C-ABI call (a replayed graph cannot be bracketed per kernel; kernels and shapes are identical)
    kernel_times, spans = {}, []
    if not args.no_kernel_timing:
        mapper.overlap_smooth = False           # one stream: an event pair must bracket its own kernel only
        ops.timer.arm(kernels=True)             # + the library's own event pair around every kernel launch
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        kernel_times = ops.timer.disarm()
        spans = ops.timer.kernel_spans
        mapper.overlap_smooth = overlap
    ms_per_step = elapsed * 1e3 / args.steps
    value = job_rays * S / (ms_per_step / 1e3)

    if ctx.rank != 0:
        return
    roofline = None
    breakdown, per_kernel = {}, []
    pmc = None
    try:        # HBM bytes per launch / per step from the committed rocprofv3 --pmc passes of this same command (profiles/)
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if pmc.get("workload") != args.workload:
            pmc = None
    except Exception:
        pass
    if kernel_times:
        tot_ms = sum(v[1] for v in kernel_times.values()) or 1.0
        for k, (calls, ms, units) in sorted(kernel_times.items(), key=lambda kv: -kv[1][1]):
            breakdown[k] = {"calls_per_step": calls / args.steps, "ms_per_step": ms / args.steps, "share": ms / tot_ms}
        per_kernel = kernel_rooflines(spans, wl, args.steps, pmc)
        modelled = [r for r in per_kernel if "frac" in r]
        if modelled:
            roofline = dict(modelled[0])            # the kernel with the largest summed duration in the timed steps
            if pmc:
                roofline["traffic_source"] = pmc["source"]
    # whole-iteration rooflines of SURVEY 8d (per ray-sample and per GPU): HBM with the algorithmic 1024 B gathered + 2048 B
    # scattered + the 12-byte point, fp32 MFMA with 3 x the forward flops of the four render networks
    nn_, nl_ = wl["nn"], wl["nl"]
    macs_ = lambda n_in, n_out: n_in * nn_ + (nl_ - 1) * nn_ * nn_ + nn_ * n_out
    flops_sample = 3 * 2 * (2 * macs_(80, 33) + macs_(112, 3) + macs_(112, 8))       # fwd + dX + dW of the four render networks
    mlp_peak = F16_MFMA_PEAK_TFLOPS / (1 if wl.get("mlp_dtype") == "fp16" else SPLIT_PRODUCTS)
    bytes_sample = 3 * 16 * 8 * 2 * 4 + 12
    per_gpu = value / ctx.world_size
    iteration_roofline = {
        "hbm": {"bytes_per_ray_sample": bytes_sample, "achieved": per_gpu * bytes_sample / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": per_gpu * bytes_sample / 1e9 / HBM_PEAK_GBS},
        "mfma": {"flops_per_ray_sample": flops_sample, "achieved": per_gpu * flops_sample / 1e12,
                 "peak": mlp_peak, "unit": "TFLOP/s", "frac": per_gpu * flops_sample / 1e12 / mlp_peak},
        "traffic": pmc.get("hbm_bytes_per_step") if pmc else None,
        "algorithmic_bytes_per_step": bytes_sample * n_rays * S,
        "note": "per GPU, whole mapping iteration incl. the smoothness lattice, losses and Adam (not counted in the numerators); "
                "traffic = sum of the PMC FETCH+WRITE bytes of every kernel of one iteration (profiles/pmc_traffic.json)"}
    out = {
        "metric": "ray-samples/s", "value": value, "unit": "ray-samples/s", "n_gpus": ctx.world_size,
        "rccl_ranks": ctx.group_size(), "dist_backend": ctx.backend_name(),
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step, "step_ms_spread": spread, "higher_is_better": True,
        "scaling": "strong" if union else "weak", "vs_baseline": None, "dtype": ("f16 activations and MFMA operands, f32 accumulate, loss scale 128 (MLPs: half rows); f32 elsewhere" if wl.get("mlp_dtype") == "fp16" else
                  "f32 (MLP products as 3 x f16 split-operand MFMA / 6 x bf16 for weight gradients, f32 accumulate: error <= fp32 fma chain)"), "data": "synthetic", "hip_graph": graphed, "streams": 2 if overlap else 1, "launch_mode": "hipGraph replay" if graphed else "eager",
        "step_driver": "fixed launch sequence (fused_step.MapStep)" if getattr(mapper, "map_step", None) is not None else "torch.autograd",
        "launch_trial_ms": trial, "final_loss": final_loss,
        "config": {"workload": args.workload + ": " + wl["desc"] + (" -- with --stem-features: the 2-D branch (feature_matching + "
                                "Decoder.merge on 3 reference views per target frame) runs inside every iteration" if args.stem_features else ""), "rays_per_gpu": job_rays // ctx.world_size, "samples_per_ray": S,
                   "global_rays": job_rays,
                   "parallelism": f"dp{ctx.world_size} (" + ("strong: union batch -- rank slices of ONE shared-seed batch of "
                                                              f"{n_rays} rays and of the lattice" if union else
                                                              f"weak: {n_rays} rays per rank, one batch and one lattice per rank") + ")"},
        "roofline": roofline,
        "iteration_roofline": iteration_roofline,
        "kernel_rooflines": per_kernel,
        "kernel_breakdown": breakdown,
    }
    # secondary line: forward-only full-image render of one 640x480 frame (frame_vis path), rays/s
    try:
        if args.no_render_forward:
            raise RuntimeError("skipped (--no-render-forward)")
        mapper.static_shapes = False
        torch.cuda.synchronize()
        rf = lambda: mapper.render_frame(frames["gt_color"][0], frames["gt_depth"][0], frames["gt_label"][0],
                                         frames["est_c2w"][0], n_pts_batch=65536)
        rf()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            rf()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 3
        out["render_forward"] = {"rays_per_s": cam["H"] * cam["W"] / dt, "ray_samples_per_s": cam["H"] * cam["W"] * S / dt,
                                 "ms_per_frame": dt * 1e3, "what": "full 640x480 frame, forward only, 65536-ray chunks"}
    except Exception as e:
        out["render_forward"] = {"error": f"{type(e).__name__}: {e}"}
    # secondary line: the tracking optimise step (slams/tracking.py:313-340), 500 rays, coarse-only render, pose-only Adam
    try:
        if args.no_render_forward:
            raise RuntimeError("skipped (--no-render-forward)")
        from dns_slam_amd.tracking import Tracker
        tcfg = dict(cfg)
        tracker = Tracker(tcfg, mapper.decoder, bound, cam, device=device)
        cur = {"gt_color": frames["gt_color"][1], "gt_depth": frames["gt_depth"][1], "gt_label": frames["gt_label"][1]}
        n_it = tcfg["tracking"]["n_iters"]
        res = {}
        for mode in ("eager", "graph"):
            tracker.track_frame(cur, frames["est_c2w"][1], n_iters=n_it, fused=True, graph=(mode == "graph"))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            tracker.track_frame(cur, frames["est_c2w"][1], n_iters=n_it, fused=True, graph=(mode == "graph"))
            torch.cuda.synchronize()
            res[mode] = (time.perf_counter() - t1) * 1e3 / n_it
        # the same loop as a fixed launch sequence (fused_step.TrackStep), eager and as one captured iteration replayed n_it times
        tracker.use_track_step = True
        for fused in (False, True):
            tracker.use_fused_kernel = fused
            for mode in ("eager", "graph"):
                tracker.track_frame(cur, frames["est_c2w"][1], n_iters=n_it, graph=(mode == "graph"))
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    tracker.track_frame(cur, frames["est_c2w"][1], n_iters=n_it, graph=(mode == "graph"))
                torch.cuda.synchronize()
                res[("fused_" if fused else "step_") + mode] = (time.perf_counter() - t1) * 1e3 / (3 * n_it)
        out["tracking"] = {"ms_per_iter_eager": res["eager"], "ms_per_iter_graph_incl_capture": res["graph"],
                           "ms_per_iter_track_step_eager": res["step_eager"],
                           "ms_per_iter_track_step_graph": res["step_graph"],       # the tracker keeps its TrackStep: capture paid once, by the first frame
                           # round 5: the iteration as ONE kernel + a pose kernel (2 launches; csrc/track_fused.inc), incl. the frame's
                           # up-front draws and dns_track_fused_begin
                           "ms_per_iter_fused_eager": res["fused_eager"], "ms_per_iter_fused_graph": res["fused_graph"],
                           "launches_per_iter_fused": 2,
                           "rays": tcfg["tracking"]["n_pixels"], "samples_per_ray": S, "iters_per_frame": n_it}
    except Exception as e:
        out["tracking"] = {"error": f"{type(e).__name__}: {e}"}
    # secondary line: the same shapes with the networks in the REFERENCE's own precision (tcnn: f16 activations / weights, loss
    # scale 128 -- models/decoder.py:58-64,94; half rows, ABI v12).  Not the headline: `value` above keeps the fp32-grade networks
    # that BASELINE's 1e-4 parity asks for.  One GPU, the default workload only; skipped with --no-render-forward.
    skip16 = args.no_render_forward or ctx.world_size != 1 or args.workload != "cfg2" or args.stem_features or args.autograd_step
    try:
        if skip16:
            raise StopIteration
        wl16 = WORKLOADS["cfg2_fp16"]
        _, _, _, _, mapper16, step16 = build(wl16, device, seed=1000, dist_ctx=ctx, overlap=True, graph=False)
        for _ in range(50):
            step16()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n16 = 200
        for _ in range(n16):
            step16()
        torch.cuda.synchronize()
        dt16 = (time.perf_counter() - t1) / n16
        l16 = float(step16().detach())
        if not math.isfinite(l16):
            raise RuntimeError("non-finite loss")
        out["reference_precision"] = {"workload": "cfg2_fp16", "ms_per_step": dt16 * 1e3, "ray_samples_per_s": 4 * sum(wl16["rays"]) * S / dt16,
                                      "dtype": "f16 activations and MFMA operands, f32 accumulate, loss scale 128 (half rows, ABI v12)",
                                      "half_rows": bool(mapper16.map_step.half), "steps": n16, "final_loss": l16,
                                      "what": "cfg2's shapes, the MLPs in tcnn's own precision; eager, two streams; a secondary line"}
        del mapper16, step16
    except StopIteration:
        pass                                                  # (not this run's leg: more than one rank / another workload / --no-render-forward)
    except Exception as e:
        out["reference_precision"] = {"error": f"{type(e).__name__}: {e}"}
    if args.loop > 0 and ctx.world_size == 1:
        try:
            mapper.overlap_smooth, mapper.prefetch_draws = True, True
            mapper._pending_draws = None
            out["slam_loop"] = slam_loop(cfg, bound, cam, frames, mapper, step, device, args.loop)
        except Exception as e:
            out["slam_loop"] = {"error": f"{type(e).__name__}: {e}"}
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(wl, cfg, bound, cam, frames)
        except Exception as e:                               # never lose the GPU line to a host-side problem
            out["cpu_baseline"] = {"value": None, "unit": "ray-samples/s", "cores": host_cores(), "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
