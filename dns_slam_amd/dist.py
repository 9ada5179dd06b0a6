"""Ray-batch data parallelism over the GPUs of one node: one process per GPU, parameters
replicated, every rank renders its own slice of the ray batch, the gradients are all-reduced once per iteration
(RCCL over xGMI via torch.distributed backend "nccl"; "gloo" on CPU for tests).

Two modes (``DistCtx.mode``):
* ``"weak"``  -- every rank draws its OWN batch (per-rank seed): the global batch is W times the configured one.  Per
  iteration: one MAX over K floats (the per-frame max sampled depth, before sampling), one SUM over the 16 loss numerators /
  counts (before the backward), the gradient all-reduce.  Every rank evaluates its own smoothness lattice at weight 1/W.
* ``"union"`` -- SURVEY 8e's partitioning: all ranks draw the SAME pixel list / jitter / lattice offsets (shared seed, same
  generator calls), rank r renders rays [r n/W, (r+1) n/W) of every frame's list and x-planes [r 63/W, (r+1) 63/W) of the
  smoothness lattice (+ one halo plane): the W ranks together compute exactly the one-GPU iteration on the configured batch
  (strong scaling).  The MAX needs no collective (every rank holds the whole list); per iteration: the 16-float SUM and the
  gradient all-reduce.

Gradient exchange (``GradBuckets``): the parameters' ``.grad`` tensors are VIEWS into persistent flat fp32 buckets (no
per-iteration allocation, flattening or copy-back).  The bucket of the colour / logit / fine-decoder gradients -- all
produced on the main stream, complete when the ray branch's MLP backward ends -- is launched asynchronously from the
post-accumulate hook of its last parameter and travels while the hash-grid scatter, the pose backward and the lattice branch
still run; the bucket holding the table, the coarse network (both ALSO fed by the lattice branch, which runs on a second
stream) and the poses is launched from ``finish()``, after ``backward()`` has returned and the autograd engine has joined
every leaf stream with the caller's (a collective is ordered only against the stream it is launched from).

The reference has no live distributed path (its NCCL helpers utils/common.py:79-162 are dead code); this is the
north_star's addition.  Exactness w.r.t. one GPU rendering the union batch: the fused loss kernel (csrc/losses.hip) first produces the 16
numerators / counts of the rank's rays; ``allreduce_sums`` makes them global BEFORE they are normalised, so every
masked mean (depth over d > 0, rays inside the box) uses the global denominator and the fs/opacity branch flag
(utils/common.py:794) is evaluated on the global counts.  Each rank then back-propagates local numerators over
global denominators and the gradients are SUMMED; the smoothness term (each rank draws its own random lattice unless
the caller passes the same offsets) is weighted 1/W, i.e. the ranks' lattices are averaged.
Bucket: all gradients are flattened into one fp32 buffer -- 6.8 MB (T=2^16) .. 59 MB (T=2^20) + <1 MB of MLPs
+ 7 floats per frame -- so the collective is one large message; xGMI is point-to-point (7 links x ~153 GB/s), a
ring all-reduce of M bytes moves 2*(7/8)*M over each link: 6.8 MB -> ~80 us, 59 MB -> ~0.7 ms.
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def shard_range(n: int, world: int, rank: int):
    """[a, b): rank's contiguous share of n items, sizes differing by at most one (the first n % world ranks get the extra)."""
    q, r = divmod(n, world)
    a = rank * q + min(rank, r)
    return a, a + q + (1 if rank < r else 0)


def union_point_labels(labels_union: torch.Tensor, n_frames: int, n_per_frame: int, a: int, b: int, n_samples: int) -> torch.Tensor:
    """Fine-decoder routing labels of a union-batch shard under the reference's tiled layout (slams/mapping.py:613):
    point k of the WHOLE batch (ray-major, N = n_frames * n_per_frame rays) is routed by labels[k mod N].  The shard holds rays
    [a, b) of every frame's list; returns the labels of its points, ray-major, [n_frames * (b - a) * n_samples]."""
    dev = labels_union.device
    N = n_frames * n_per_frame
    gid = (torch.arange(n_frames, device=dev)[:, None] * n_per_frame + torch.arange(a, b, device=dev)[None]).reshape(-1)
    k = gid[:, None] * n_samples + torch.arange(n_samples, device=dev)[None]
    return labels_union.reshape(-1)[k % N].reshape(-1)


class GradBuckets:
    """Persistent flat gradient buckets with hook-launched asynchronous all-reduces (see the module docstring).

    ``groups``: list of parameter lists, in LAUNCH order (collectives must be issued in the same order on every rank: bucket
    k is launched only after buckets < k, whatever order autograd finishes them in).  Usage per iteration::

        buckets.zero()            # instead of optimizer.zero_grad(): .grad stays a view of the bucket
        loss.backward()           # hooks launch bucket k once its last parameter has accumulated
        buckets.finish()          # launch what no hook launched (parameters without a gradient this iteration), wait
        optimizer.step()
    """

    def __init__(self, ctx: "DistCtx", groups, hook_launch=None):
        """``hook_launch[k]``: may bucket k's all-reduce be launched from the autograd hook of its last parameter?  NCCL/RCCL
        orders a collective only against the stream that is current where it is launched; a hook runs on the autograd thread
        under the stream of THAT parameter's AccumulateGrad node.  That is safe exactly when every gradient of the bucket is
        produced on one stream (the colour / logit / fine-decoder bucket: all main-stream).  A bucket whose parameters also
        receive gradient from a second stream (table and coarse network: the lattice branch) must be launched from
        ``finish()``, after ``backward()`` has returned -- the engine has then joined every leaf stream with the caller's.
        Default: every bucket hook-launched (single-stream callers)."""
        self.ctx = ctx
        groups = [[p for p in g if p.requires_grad and p.numel() > 0] for g in groups]
        if hook_launch is not None and len(hook_launch) != len(groups):
            raise ValueError("GradBuckets: hook_launch needs one flag per group")
        flags = [True] * len(groups) if hook_launch is None else [bool(h) for h in hook_launch]
        kept = [(g, h) for g, h in zip(groups, flags) if g]             # a group can end up empty (all frozen): drop its FLAG with it
        self.groups = [g for g, _ in kept]
        self._hook_launch = None if hook_launch is None else [h for _, h in kept]
        self.flat, self._pending, self._works, self._next = [], [], [], 0
        self._hooks = []
        for k, g in enumerate(self.groups):
            flat = torch.zeros(sum(p.numel() for p in g), device=g[0].device, dtype=torch.float32)
            o = 0
            for p in g:
                p.grad = flat[o:o + p.numel()].view_as(p)
                o += p.numel()
                if ctx.enabled:
                    self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, k=k: self._ready(k)))
            self.flat.append(flat)
            self._pending.append(len(g))
        self._counts = list(self._pending)

    def nbytes(self):
        return [f.numel() * 4 for f in self.flat]

    def zero(self):
        for f in self.flat:
            f.zero_()
        self._pending = list(self._counts)
        self._works, self._next = [], 0

    def _launch_through(self, k):
        while self._next <= k:
            self._works.append(dist.all_reduce(self.flat[self._next], op=dist.ReduceOp.SUM, group=self.ctx.group, async_op=True))
            self._next += 1

    def _ready(self, k):
        self._pending[k] -= 1
        # launch in bucket order: bucket k goes once it AND every earlier bucket are complete
        while self._next < len(self.flat) and self._pending[self._next] <= 0 and \
                (self._hook_launch is None or self._hook_launch[self._next]):
            self._launch_through(self._next)

    def finish(self):
        if not self.ctx.enabled:
            return
        self._launch_through(len(self.flat) - 1)
        for w in self._works:
            w.wait()
        self._works = []

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


class DistCtx:
    def __init__(self, world_size: int = 1, rank: int = 0, group=None, mode: str = "weak", force: bool = False):
        """``force``: issue every collective even with world_size == 1 (a one-rank process group: the collectives are
        identities, but they run through the backend -- tests/test_dist.py drives RCCL's stream ordering on one GPU that way)."""
        if mode not in ("weak", "union"):
            raise ValueError(f"DistCtx mode '{mode}' (weak | union)")
        self.world_size, self.rank, self.group, self.mode = world_size, rank, group, mode
        self.force = bool(force)
        self._flat_cache = {}

    @property
    def union(self):
        return self.enabled and self.mode == "union"

    def shard(self, n: int):
        return shard_range(n, self.world_size, self.rank)

    def make_buckets(self, groups, hook_launch=None) -> GradBuckets:
        return GradBuckets(self, groups, hook_launch)

    @property
    def enabled(self):
        return self.world_size > 1 or self.force

    def group_size(self) -> int:
        """Ranks of the process group the collectives run in, as the backend reports it (1 without a group): what ``bench.py``
        prints as ``rccl_ranks`` next to ``n_gpus``."""
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def backend_name(self) -> str:
        return str(dist.get_backend(self.group)) if (dist.is_available() and dist.is_initialized()) else "none"

    # ---- loss fix-ups -----------------------------------------------------------------------
    def global_mean_scale(self, local_count: torch.Tensor) -> torch.Tensor:
        """Factor s_r = W * count_r / sum_r(count_r) for a masked mean whose denominator is ``local_count``."""
        if not self.enabled:
            return torch.ones((), device=local_count.device, dtype=torch.float32)
        tot = local_count.detach().clone().float()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=self.group)
        return self.world_size * local_count.detach().float() / tot

    def global_any(self, local_flag: torch.Tensor) -> torch.Tensor:
        if not self.enabled:
            return local_flag
        f = local_flag.detach().clone().float()
        dist.all_reduce(f, op=dist.ReduceOp.MAX, group=self.group)
        return f

    def allreduce_sums(self, sums: torch.Tensor) -> None:
        """In-place SUM of the fused loss kernel's numerators / counts (16 floats): every rank then normalises by the
        GLOBAL denominators and evaluates the fs/opacity branch flag on the GLOBAL counts."""
        if self.enabled:
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.group)

    def allreduce_max(self, t: torch.Tensor) -> None:
        """In-place MAX (the per-frame max sampled depth that sample_along_rays uses, utils/common.py:581,591)."""
        if self.enabled:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)

    # ---- gradient exchange ------------------------------------------------------------------
    def allreduce_grads(self, params: Iterable[torch.Tensor]) -> None:
        """SUM the gradients of ``params`` over the ranks with ONE all-reduce of a flat fp32 bucket.  Each rank's loss
        is (local numerators) / (global denominators), so the sum of the per-rank gradients IS the gradient of the loss
        of the union batch.  Parameters without a gradient on this rank (e.g. a frozen pose) contribute zeros."""
        if not self.enabled:
            return
        ps: List[torch.Tensor] = [p for p in params if p.requires_grad]
        if not ps:
            return
        n = sum(p.numel() for p in ps)
        key = (n, str(ps[0].device))
        flat = self._flat_cache.get(key)                  # persistent: no allocation per iteration
        if flat is None:
            flat = self._flat_cache[key] = torch.empty(n, device=ps[0].device, dtype=torch.float32)
        o = 0
        for p in ps:
            if p.grad is not None:
                flat[o:o + p.numel()].copy_(p.grad.reshape(-1))
            else:
                flat[o:o + p.numel()].zero_()
            o += p.numel()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        o = 0
        for p in ps:
            g = flat[o:o + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            o += p.numel()

    def barrier(self):
        if self.enabled:
            dist.barrier(group=self.group)

    def max_over_ranks(self, value: float, device) -> float:
        if not self.enabled:
            return value
        t = torch.tensor([value], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())


def init_from_env(backend: Optional[str] = None, mode: str = "weak", force: bool = False) -> DistCtx:
    """One process per GPU, launched by ``torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    ``force``: with WORLD_SIZE 1 still create the (one-rank) process group and return a context that issues every collective
    through it (``DistCtx.force``) -- RCCL's initialisation and stream semantics exercised on a single GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not force:
        return DistCtx(mode=mode)
    if world <= 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=0, world_size=1)
        return DistCtx(1, 0, mode=mode, force=True)
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:
        backend = os.environ.get("DNS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return DistCtx(world, rank, mode=mode)
