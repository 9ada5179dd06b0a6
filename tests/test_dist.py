"""Multi-GPU path (dns_slam_amd/dist.py): world_size-2 tests.

CPU (gloo): the flat-bucket gradient all-reduce and the loss-sum all-reduce give, for per-rank losses of the form
(local numerators) / (global denominators), exactly the gradient of the union batch -- including a masked mean whose
per-rank counts differ.  GPU (gloo over CUDA tensors, both ranks on cuda:0): the real HIP mapping iteration on two ray
shards equals the one-process iteration on the union batch (per_ray label layout; 1e-4)."""
import os
import socket

import pytest
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


def _cpu_worker(rank, world, port, q):
    ctx = _init(rank, world, port)
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(64, 5, generator=g), torch.randn(64, generator=g)
    mask = (torch.rand(64, generator=g) < 0.6).float()
    w = torch.randn(5, generator=g).requires_grad_(True)
    b = torch.zeros(()).requires_grad_(True)
    frozen = torch.zeros(3).requires_grad_(True)        # no gradient on this rank: must contribute zeros
    sl = slice(rank * 32, (rank + 1) * 32)              # uneven mask counts per shard
    loss = _toy_loss(w, b, x[sl], y[sl], mask[sl], ctx)
    loss.backward()
    ctx.allreduce_grads([w, b, frozen])
    t = ctx.max_over_ranks(float(rank), "cpu")
    ctx.barrier()
    if rank == 0:
        q.put((w.grad.numpy().copy(), b.grad.numpy().copy(), frozen.grad.numpy().copy(), t))   # numpy: no fd passing
    dist.destroy_process_group()


def test_gloo_world2_sum_of_shard_grads_is_union_grad():
    port = _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_cpu_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    gw, gb, gf, t = q.get(timeout=120)
    gw, gb, gf = torch.from_numpy(gw), torch.from_numpy(gb), torch.from_numpy(gf)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(64, 5, generator=g), torch.randn(64, generator=g)
    mask = (torch.rand(64, generator=g) < 0.6).float()
    w = torch.randn(5, generator=g).requires_grad_(True)
    b = torch.zeros(()).requires_grad_(True)
    _toy_loss(w, b, x, y, mask, None).backward()
    assert mask[:32].sum() != mask[32:].sum()           # the test exercises unequal denominators
    torch.testing.assert_close(gw, w.grad, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(gb, b.grad, rtol=1e-6, atol=1e-7)
    assert torch.count_nonzero(gf) == 0 and t == 1.0


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
def _gpu_setup(dev):
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
    mapper = Mapper(cfg, dec, bound, cam, device=dev, label_layout="per_ray")
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


def _gpu_worker(rank, world, port, q):
    ctx = _init(rank, world, port)
    dev = "cuda:0"
    frames, dec, mapper, ql, Tl, prep, pix, jit, u, npf = _gpu_setup(dev)
    mapper.dist = ctx
    half = npf // 2
    shard = torch.cat([pix[f * npf + rank * half: f * npf + (rank + 1) * half] for f in range(4)])
    s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=shard, jitter=jit)
    loss, _ = mapper.iteration_loss(s, smooth=True, u_offset=u[0], u_jitter=u[1])
    loss.backward()
    params = list(dec.parameters()) + [mapper.fine_decoders.pool] + ql + Tl
    ctx.allreduce_grads([p for p in params if p.numel() > 0])
    torch.cuda.synchronize()
    if rank == 0:
        q.put([g.numpy() for g in _grads(dec, mapper, ql, Tl)])      # numpy: plain bytes, no fd hand-shake with a dying child
    ctx.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_shards_equal_union_batch_on_gpu():
    port = _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    sharded = [torch.from_numpy(a) for a in q.get(timeout=300)]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    frames, dec, mapper, ql, Tl, prep, pix, jit, u, npf = _gpu_setup("cuda:0")
    half = npf // 2
    # the per-frame max(gt_depth) of sample_along_rays is all-reduced (MAX) across the shards, so z is identical
    union = torch.cat([pix[f * npf: f * npf + 2 * half] for f in range(4)])
    s = mapper.get_target_samples(frames, ql, Tl, prep=prep, pix_idx=union, jitter=jit)
    loss, _ = mapper.iteration_loss(s, smooth=True, u_offset=u[0], u_jitter=u[1])
    loss.backward()
    ref = _grads(dec, mapper, ql, Tl)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import rel_err
    errs = [rel_err(a, b) for a, b in zip(sharded, ref)]
    assert max(errs) < 2e-4, errs          # fp32 sums in a different order; everything else is exact
