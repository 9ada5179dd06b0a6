"""Multi-GPU path (dns_slam_amd/dist.py): world_size 2, 4 and 8.

CPU (gloo): the gradient exchange -- the flat all-reduce and the persistent two-bucket form whose all-reduces are launched
from post-accumulate hooks -- and the loss-sum all-reduce give, for per-rank losses of the form (local numerators) /
(global denominators), exactly the gradient of the union batch, including a masked mean whose per-rank counts differ, a
parameter without a gradient on some rank, and buckets that complete in the opposite of their launch order; the shard
arithmetic of the union-batch mode (ray ranges, lattice slabs with halo planes, tiled routing labels).
GPU (gloo over CUDA tensors, all ranks on cuda:0): the real HIP mapping iteration on W ray shards equals the one-process
iteration on the whole batch, in weak mode (explicit shards, per_ray labels) and in union-batch mode (shared draws, rank
slices, sharded lattice, reference_tiled labels), 1e-4."""
import os
import socket

import pytest
import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = str(rank), str(world), str(rank)
    from dns_slam_amd import dist as dd
    return dd.init_from_env(backend="gloo")


# ----------------------------------------------------------------------------------------------- CPU plumbing
def _toy_loss(w, b, x, y, mask, ctx):
    """Masked-mean regression loss in the fused-loss contract: sums -> (all-reduce) -> normalise."""
    pred = x @ w + b
    num = (((pred - y) ** 2) * mask).sum()
    sums = torch.stack([num.detach(), mask.sum()])
    if ctx is not None:
        ctx.allreduce_sums(sums)
    return num / sums[1]                      # local numerator over GLOBAL denominator


def _toy_data():
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(64, 5, generator=g), torch.randn(64, generator=g)
    mask = (torch.rand(64, generator=g) < 0.6).float()
    w = torch.randn(5, generator=g).requires_grad_(True)
    b = torch.zeros(()).requires_grad_(True)
    return x, y, mask, w, b


def _cpu_worker(rank, world, port, q, buckets):
    from dns_slam_amd.dist import shard_range
    ctx = _init(rank, world, port)
    x, y, mask, w, b = _toy_data()
    frozen = torch.zeros(3).requires_grad_(True)        # no gradient on any rank: must contribute zeros
    rare = torch.ones(2).requires_grad_(True)           # gradient on rank 0 only
    a, e = shard_range(64, world, rank)                 # uneven mask counts per shard
    sl = slice(a, e)
    ptrs = None
    if buckets:
        # launch order [w, frozen] then [b, rare]; autograd finishes b (the last op of the forward) BEFORE w: the second
        # bucket is complete first and must wait for the first; `frozen` never fires its hook: finish() launches the rest
        # buckets == "late": only the first bucket may be launched from its hook, the second goes out from finish() -- what
        # bench.py uses for the bucket whose gradients are produced on two streams (dist.py GradBuckets)
        bk = ctx.make_buckets([[w, frozen], [b, rare]], hook_launch=[True, False] if buckets == "late" else None)
        ptrs = [p.grad.data_ptr() for p in (w, frozen, b, rare)]
    for it in range(2):                                 # two iterations: the persistent views survive and are re-zeroed
        if buckets:
            bk.zero()
        else:
            for p in (w, b, frozen, rare):
                p.grad = None
        loss = _toy_loss(w, b, x[sl], y[sl], mask[sl], ctx)
        if rank == 0:
            loss = loss + (rare * torch.tensor([2.0, -3.0])).sum()
        loss.backward()
        if buckets:
            bk.finish()
        else:
            ctx.allreduce_grads([w, b, frozen, rare])
    if buckets:
        assert ptrs == [p.grad.data_ptr() for p in (w, frozen, b, rare)], "a .grad left its bucket"
        assert bk.nbytes() == [8 * 4, 3 * 4]
    t = ctx.max_over_ranks(float(rank), "cpu")
    ctx.barrier()
    if rank == 0:
        q.put((w.grad.numpy().copy(), b.grad.numpy().copy(), frozen.grad.numpy().copy(), rare.grad.numpy().copy(), t))   # numpy: no fd passing
    dist.destroy_process_group()


@pytest.mark.parametrize("world,buckets", [(2, False), (2, True), (2, "late"), (4, False), (4, True), (8, "late"), (8, True)])
def test_gloo_sum_of_shard_grads_is_union_grad(world, buckets):
    port = _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_cpu_worker, args=(r, world, port, q, buckets)) for r in range(world)]
    [p.start() for p in procs]
    gw, gb, gf, gr, t = q.get(timeout=180)
    gw, gb, gf, gr = [torch.from_numpy(v) for v in (gw, gb, gf, gr)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    x, y, mask, w, b = _toy_data()
    _toy_loss(w, b, x, y, mask, None).backward()
    assert len({float(mask[i * 64 // world:(i + 1) * 64 // world].sum()) for i in range(world)}) > 1   # unequal denominators
    torch.testing.assert_close(gw, w.grad, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(gb, b.grad, rtol=1e-6, atol=1e-7)
    assert torch.count_nonzero(gf) == 0 and t == float(world - 1)
    assert torch.equal(gr, torch.tensor([2.0, -3.0]))


def test_union_mode_shard_arithmetic():
    """shard_range covers [0, n) without overlap for ragged n; the lattice slabs (+ halo plane) account for every
    total-variation term of the cube exactly once; the tiled routing labels of a shard are the whole batch's tiling
    (slams/mapping.py:613) restricted to the shard's points."""
    from dns_slam_amd.dist import shard_range, union_point_labels
    for n, W in ((63, 4), (63, 8), (1024, 3), (5, 4)):
        cuts = [shard_range(n, W, r) for r in range(W)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
        assert max(b - a for a, b in cuts) - min(b - a for a, b in cuts) <= 1
    # TV decomposition (the rule csrc/misc.hip implements for nx / halo), restated with torch on a random cube
    g = torch.Generator().manual_seed(1)
    n = 11
    occ = torch.randn(n, n, n, generator=g, dtype=torch.float64)
    tv = lambda o: ((o[1:] - o[:-1]) ** 2).sum() + ((o[:, 1:] - o[:, :-1]) ** 2).sum() + ((o[:, :, 1:] - o[:, :, :-1]) ** 2).sum()
    for W in (2, 3, 4, 8):
        tot = 0.0
        for r in range(W):
            a, b = shard_range(n, W, r)
            hi = min(b + 1, n)
            slab, halo = occ[a:hi], hi > b
            own = slab[:-1] if halo else slab
            tot = tot + ((slab[1:] - slab[:-1]) ** 2).sum() + ((own[:, 1:] - own[:, :-1]) ** 2).sum() + ((own[:, :, 1:] - own[:, :, :-1]) ** 2).sum()
        assert abs(float(tot - tv(occ))) <= 1e-12 * float(tv(occ))
    K, npf, S = 3, 10, 4
    lab = torch.randint(0, 8, (K * npf,), generator=g)
    tiled = lab.repeat(1, S).flatten(0, 1)                              # the reference's expression on the whole batch
    for W in (2, 4, 8):
        for r in range(W):
            a, b = shard_range(npf, W, r)
            got = union_point_labels(lab, K, npf, a, b, S)
            rays = torch.cat([torch.arange(f * npf + a, f * npf + b) for f in range(K)])
            want = tiled.reshape(K * npf, S)[rays].reshape(-1)          # point (ray, s) of the whole batch is tiled[ray * S + s]
            assert torch.equal(got, want)


def test_single_process_ctx_is_a_noop():
    from dns_slam_amd.dist import DistCtx
    ctx = DistCtx()
    p = torch.ones(3, requires_grad=True)
    p.grad = torch.full((3,), 2.0)
    ctx.allreduce_grads([p])
    s = torch.arange(16.0)
    ctx.allreduce_sums(s)
    assert not ctx.enabled and torch.equal(p.grad, torch.full((3,), 2.0)) and torch.equal(s, torch.arange(16.0))
    assert ctx.max_over_ranks(3.5, "cpu") == 3.5


# ----------------------------------------------------------------------------------------------- GPU: real path
def _gpu_setup(dev, layout="per_ray"):
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.mapping import Mapper
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import randomise_
    cam = synthetic.camera(H=60, W=80, fx=60.0, fy=60.0)
    bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=0)
    cfg = synthetic.default_cfg(n_pixels=480, n_samples_ray=32, n_surface_ray=15, hash_size=14, voxel_size=0.08, smooth_pts=10)
    dec = Decoder(cfg["model"], bound, n_class=8).to(dev)
    mapper = Mapper(cfg, dec, bound, cam, device=dev, label_layout=layout)
    mapper.static_shapes = True
    mapper.set_decoder(frames)
    randomise_(dec, 11)
    with torch.no_grad():
        dec.pe_fn.grid_fn.params.mul_(2000.0)
    randomise_([mapper.fine_decoders.pool], 12)
    mapper.is_BA = True
    _, ql, Tl = mapper.set_optimizer(frames)
    prep = mapper.prepare_frames(frames)
    g = torch.Generator().manual_seed(5)
    npf = prep["n1"] + prep["n2"]
    pix = torch.randint(60 * 80, (4 * npf,), generator=g).to(dev)
    t = torch.rand(15, generator=g)
    t[8] = 0.5
    jit = (t.to(dev), torch.rand(15, generator=g).to(dev))
    u = (torch.rand(3, generator=g), torch.rand((1, 1, 1, 3), generator=g))
    return frames, dec, mapper, ql, Tl, prep, pix, jit, u, npf


def _grads(dec, mapper, ql, Tl):
    out = [dec.pe_fn.grid_fn.params.grad, dec.coarse_fn.decoder.params.grad, dec.out_fn.color_decoder.params.grad,
           dec.out_fn.logit_decoder.params.grad, mapper.fine_decoders.pool.grad] + [q.grad for q in ql[1:]] + [t.grad for t in Tl[1:]]
    return [g.detach().cpu().clone() for g in out]


def _gpu_worker(rank, world, port, q, mode):
    ctx = _init(rank, world, port)
    ctx.mode = mode
    dev = "cuda:0"
    frames, dec, mapper, ql, Tl, prep, pix, jit, u, npf = _gpu_setup(dev, "reference_tiled" if mode == "union" else "per_ray")
    mapper.dist = ctx
    part = npf // world
    params = [p for p in list(dec.parameters()) + [mapper.fine_decoders.pool] + ql + Tl if p.numel() > 0]
    if mode == "union":
        # every rank is handed the WHOLE list (what a shared seed gives); the mapper takes its slice, its lattice slab, the
        # tiled routing labels of the whole batch; gradients travel in two persistent buckets launched from hooks
        early = [dec.out_fn.color_decoder.params, dec.out_fn.logit_decoder.params, mapper.fine_decoders.pool]
        late = [p for p in params if all(p is not e for e in early)]
        bk = ctx.make_buckets([early, late])
        bk.zero()
        whole = torch.cat([pix[f * npf: f * npf + world * part] for f in range(4)])
        s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=whole, jitter=jit)
        assert s["z_vals"].shape[0] == 4 * part
    else:
        shard = torch.cat([pix[f * npf + rank * part: f * npf + (rank + 1) * part] for f in range(4)])
        s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=shard, jitter=jit)
    loss, _ = mapper.iteration_loss(s, smooth=True, u_offset=u[0], u_jitter=u[1])
    loss.backward()
    if mode == "union":
        bk.finish()
    else:
        ctx.allreduce_grads(params)
    torch.cuda.synchronize()
    if rank == 0:
        q.put([g.numpy() for g in _grads(dec, mapper, ql, Tl)])      # numpy: plain bytes, no fd hand-shake with a dying child
    ctx.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "weak"), (2, "union"), (4, "weak"), (4, "union")])
def test_rank_shards_equal_whole_batch_on_gpu(world, mode):
    port = _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_gpu_worker, args=(r, world, port, q, mode)) for r in range(world)]
    [p.start() for p in procs]
    sharded = [torch.from_numpy(a) for a in q.get(timeout=300)]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    frames, dec, mapper, ql, Tl, prep, pix, jit, u, npf = _gpu_setup("cuda:0", "reference_tiled" if mode == "union" else "per_ray")
    part = npf // world
    # the per-frame max(gt_depth) of sample_along_rays is global in both modes (all-reduced MAX / taken from the whole list)
    whole = torch.cat([pix[f * npf: f * npf + world * part] for f in range(4)])
    s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=whole, jitter=jit)
    loss, _ = mapper.iteration_loss(s, smooth=True, u_offset=u[0], u_jitter=u[1])
    loss.backward()
    ref = _grads(dec, mapper, ql, Tl)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import rel_err
    errs = [rel_err(a, b) for a, b in zip(sharded, ref)]
    assert max(errs) < 1e-4, errs          # fp32 sums in a different order; everything else is exact


def _map_step_grads(ms):
    return [g.detach().cpu().clone() for g in (ms.g_table, ms.g_coarse, ms.g_color, ms.g_logit, ms.g_pool, ms.g_quat[4:], ms.g_trans[3:])]


def _map_step_worker(rank, world, port, q, mode="weak"):
    from dns_slam_amd.fused_step import MapStep
    ctx = _init(rank, world, port)
    ctx.mode = mode
    union = mode == "union"
    frames, dec, mapper, ql, Tl, prep, pix, jit, u, npf = _gpu_setup("cuda:0", "reference_tiled" if union else "per_ray")
    mapper.dist = ctx
    mapper.overlap_smooth = True
    part = npf // world
    g = torch.Generator().manual_seed(8)
    code = (torch.rand(4 * npf, 32 + 15, 32, generator=g) * 2 - 1).to("cuda:0") if union else None
    if union:
        # union-batch mode: every rank is handed the WHOLE list (what a shared seed gives) and takes its slice of the rays, of the
        # per-sample code and of the lattice (slab + halo plane); the reference's tiled labels route by the GLOBAL point index
        ms = MapStep(mapper, frames, ql, Tl, prep=prep, features=code)
        assert ms.union and ms.N == 4 * (ms.ray_b - ms.ray_a) and ms.Pl < (10 - 1) ** 3
        shard = pix
    else:
        mapper.rays_per_frame = (part, 0)                  # the buffers are sized for this rank's share
        prep = mapper.prepare_frames(frames)
        ms = MapStep(mapper, frames, ql, Tl, prep=prep)
        shard = torch.cat([pix[f * npf + rank * part: f * npf + (rank + 1) * part] for f in range(4)])
    r6 = torch.cat((u[0].reshape(-1), u[1].reshape(-1))).to("cuda:0")
    ms.step(draws={"pix": shard, "jitter": jit, "r6": r6})
    torch.cuda.synchronize()
    first = [g.numpy() for g in _map_step_grads(ms)] + [ms.out.cpu().numpy()]
    # then free-running steps, every rank on its own draws (per-rank seed; union mode: the SAME seed), the next step's set
    # prepared on the side stream (weak mode: its MAX all-reduce of the depth maxima is issued from there): the replicas must
    # stay bit-identical
    mapper.prefetch_draws = True
    torch.manual_seed(100 + (0 if union else rank))
    torch.cuda.manual_seed(100 + (0 if union else rank))
    for _ in range(3):
        ms.step()
    torch.cuda.synchronize()
    chk = torch.stack([p.detach().double().sum() for p in (ms.p_table, ms.p_coarse, ms.p_color, ms.p_logit, ms.p_pool, ms.Q, ms.T)])
    both = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(both, chk)
    same = all(torch.equal(both[0], b) for b in both) and bool(torch.isfinite(chk).all())
    if rank == 0:
        q.put(first + [np.array([1.0 if same else 0.0])])
    ctx.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "weak"), (2, "union"), (3, "union")])
def test_map_step_rank_shards_equal_whole_batch_on_gpu(world, mode):
    """The fixed-launch-sequence iteration (fused_step.MapStep) under data parallelism, ranks over gloo on one GPU: the
    all-reduced gradient buffer (early segment launched asynchronously after the MLP backward, late segment after the stream join)
    and the loss terms (global numerators / denominators) equal the one-process step on the whole pixel list.  weak: explicit
    shards, per_ray labels; union (SURVEY 8e's partitioning): shared draws, rank slices of the rays (ragged with 3 ranks) and of
    the per-sample code, lattice slabs + halo planes, the reference's tiled labels."""
    from dns_slam_amd.fused_step import MapStep
    port = _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_map_step_worker, args=(r, world, port, q, mode)) for r in range(world)]
    [p.start() for p in procs]
    got = [torch.from_numpy(a) for a in q.get(timeout=300)]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    union = mode == "union"
    frames, dec, mapper, ql, Tl, prep, pix, jit, u, npf = _gpu_setup("cuda:0", "reference_tiled" if union else "per_ray")
    part = npf // world
    code = None
    if union:
        g = torch.Generator().manual_seed(8)
        code = (torch.rand(4 * npf, 32 + 15, 32, generator=g) * 2 - 1).to("cuda:0")
    else:
        mapper.rays_per_frame = (world * part, 0)
        prep = mapper.prepare_frames(frames)
    ms = MapStep(mapper, frames, ql, Tl, prep=prep, features=code)
    whole = pix if union else torch.cat([pix[f * npf: f * npf + world * part] for f in range(4)])
    r6 = torch.cat((u[0].reshape(-1), u[1].reshape(-1))).to("cuda:0")
    ms.step(draws={"pix": whole, "jitter": jit, "r6": r6})
    torch.cuda.synchronize()
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import rel_err
    assert float(got[-1][0]) == 1.0, "the replicas diverged over free-running MapStep iterations"
    errs = [rel_err(a, b) for a, b in zip(got[:-2], _map_step_grads(ms))]
    assert max(errs) < 1e-4, errs
    assert rel_err(got[-2][:7], ms.out.cpu()[:7]) < 1e-5


@pytest.mark.gpu
def test_tv_slabs_sum_to_the_cube_on_gpu():
    """dns_tv_fwd / dns_tv_bwd with nx / halo: slab values and slab gradients (halo rows overlapping the next slab's first
    plane) add up to the whole lattice's value and gradient."""
    from dns_slam_amd import ops
    from dns_slam_amd.dist import shard_range
    g = torch.Generator().manual_seed(2)
    n, L = 13, 3
    lat = torch.randn(n * n * n, L, generator=g).to("cuda")
    ref_in = lat.clone().requires_grad_(True)
    ref = ops.tv_smoothness(ref_in, n, n + 1)
    ref.backward()
    for W in (2, 4, 5):
        tot, grad = 0.0, torch.zeros_like(lat)
        for r in range(W):
            a, b = shard_range(n, W, r)
            hi = min(b + 1, n)
            slab = lat[a * n * n: hi * n * n].clone().requires_grad_(True)
            v = ops.tv_smoothness(slab, n, n + 1, nx=hi - a, halo=hi > b)
            v.backward()
            tot = tot + float(v)
            grad[a * n * n: hi * n * n] += slab.grad
        assert abs(tot - float(ref)) <= 1e-5 * abs(float(ref))
        assert float((grad - ref_in.grad).abs().max()) <= 1e-5 * float(ref_in.grad.abs().max())


@pytest.mark.gpu
def test_union_mode_rank_slice_is_a_slice_of_the_whole_batch():
    """Single process, a stand-in DistCtx (no collective is needed in union mode once the draws are known to agree): rank r of
    W in union-batch mode gets exactly rays [a, b) of every frame's list -- samples, the per-sample feature code through the
    truncation mask, and the tiled routing labels of the WHOLE batch."""
    from dns_slam_amd.dist import DistCtx, shard_range
    frames, dec, mapper, ql, Tl, prep, pix, jit, u, npf = _gpu_setup("cuda:0", "reference_tiled")
    S = 32 + 15
    g = torch.Generator().manual_seed(3)
    code = (torch.rand(4 * npf, S, 32, generator=g) * 2 - 1).to("cuda:0")
    whole = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=pix, jitter=jit, features=code)
    tiled = whole["gt_label"].repeat(1, S).flatten(0, 1).reshape(4 * npf, S)          # the reference's expression (mapping.py:613)

    class Ctx(DistCtx):
        def allreduce_max(self, t):          # one process: the ranks' draws agree by construction
            pass

    for W, r in ((2, 1), (3, 0), (4, 3)):
        ctx = Ctx(W, r, mode="union")
        assert ctx.union
        mapper.dist = ctx
        mapper._union_checked = False
        part = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=pix, jitter=jit, features=code)
        a, b = shard_range(npf, W, r)
        rows = torch.cat([torch.arange(f * npf + a, f * npf + b) for f in range(4)]).to("cuda:0")
        for key in ("pts", "z_vals", "gt_depth", "gt_color", "rays_d", "features", "valid"):
            assert torch.equal(part[key], whole[key][rows]), (W, r, key)
        assert torch.equal(part["point_labels"], tiled[rows].reshape(-1)), (W, r)
    mapper.dist = None


def _nccl_one_rank_worker(port, q):
    """One process, ONE-rank RCCL group (backend "nccl"), ``DistCtx.force``: every distributed branch of ``MapStep.step`` runs
    -- the MAX of the depth maxima issued from the SIDE stream a step ahead, the 16-float SUM between dns_loss_sums and
    dns_loss_finalize on the main stream, the early gradient bucket all-reduced asynchronously from the side stream behind the
    last forked dW_in, the late bucket after the stream join, ``work.wait()`` on both before Adam.  A one-rank all-reduce is the
    identity, so the result must equal the non-distributed step: what this exercises is RCCL's initialisation and the ordering
    of its kernels against both of MapStep's streams (gloo synchronises on the host and cannot show that)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import rel_err
    from dns_slam_amd import dist as dd
    from dns_slam_amd.fused_step import MapStep
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["WORLD_SIZE"], os.environ["RANK"], os.environ["LOCAL_RANK"] = "1", "0", "0"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    ctx = dd.init_from_env(backend="nccl", force=True)
    assert ctx.enabled and ctx.world_size == 1 and dist.get_backend() == "nccl"
    out = {}
    for name, c in (("plain", None), ("nccl", ctx)):
        frames, dec, mapper, ql, Tl, prep, pix, jit, u, npf = _gpu_setup("cuda:0", "per_ray")
        mapper.dist = c
        mapper.overlap_smooth, mapper.prefetch_draws = True, False
        ms = MapStep(mapper, frames, ql, Tl, prep=prep)
        assert ms.dist_on == (c is not None)
        r6 = torch.cat((u[0].reshape(-1), u[1].reshape(-1))).to("cuda:0")
        ms.step(draws={"pix": pix, "jitter": jit, "r6": r6})
        torch.cuda.synchronize()
        first = _map_step_grads(ms) + [ms.out.cpu().clone()]
        mapper.prefetch_draws = True                       # free-running: the next step's set (and its MAX) on the side stream
        torch.manual_seed(321)
        torch.cuda.manual_seed(321)
        losses = []
        for i in range(4):
            ms.step(last=i == 3)
            losses.append(ms.out[:7].clone())
        torch.cuda.synchronize()
        params = [p.detach().cpu().clone() for p in (ms.p_table, ms.p_coarse, ms.p_color, ms.p_logit, ms.p_pool, ms.Q, ms.T)]
        out[name] = (first, torch.stack(losses).cpu(), params, float(mapper.lr), float(mapper.BA_cam_lr))
    a, b = out["plain"], out["nccl"]
    res = {"grad_err": max(rel_err(x, y) for x, y in zip(b[0], a[0])),
           "loss_err": float(((b[1] - a[1]).abs() / a[1].abs().clamp_min(1e-12)).max()),
           "finite": bool(torch.isfinite(b[1]).all())}
    # parameters after 5 Adam steps: the table scatter's float atomics make two runs of the SAME program differ in the last
    # bits, and Adam turns a last-bit gradient difference of a near-zero component into a fraction of lr: bound as in
    # test_gpu_fused_step (1e-4 of the tensor's scale + 5 % of what 5 steps can move)
    perr = []
    for k, (x, y) in enumerate(zip(b[2], a[2])):
        lr = a[4] if k >= 5 else a[3]
        perr.append(float((x - y).abs().max()) / (1e-4 * float(y.abs().max()) + 0.05 * lr * 5))
    res["param_err_over_bound"] = max(perr)
    q.put(res)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_map_step_collectives_under_rccl_one_rank_group():
    port = _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    p = ctxm.Process(target=_nccl_one_rank_worker, args=(port, q))
    p.start()
    res = q.get(timeout=600)
    p.join(120)
    assert p.exitcode == 0
    assert res["finite"] and res["grad_err"] < 1e-6 and res["loss_err"] < 1e-6, res
    assert res["param_err_over_bound"] <= 1.0, res
