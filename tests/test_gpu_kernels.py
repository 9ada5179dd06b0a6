"""GPU parity, kernel by kernel: every HIP entry point (called through the C ABI via dns_slam_amd.ops) against the
CPU oracle on the same seeded inputs, and against the committed golden vectors of the imported reference.
Integer / index work is bit-exact; floating point within 1e-4 relative (BASELINE.json), the tolerance is in the test.
"""
import os

import numpy as np
import pytest
import torch

from oracle import render_math as rm
from oracle import tcnn_ref as tr
from util import assert_close, rel_err, table_level_groups, mlp_param_groups

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from dns_slam_amd import ops
    return ops


def _t(a):
    return torch.from_numpy(np.asarray(a))


# ----------------------------------------------------------------------------------------- hash grid
@pytest.mark.parametrize("hash_size,res", [(16, 592), (20, 231), (12, 64)])
def test_hashgrid_rows_bit_exact(hash_size, res):
    ops = _ops()
    om, pm = tr.grid_meta(hash_size, res), ops.GridMeta(hash_size, res)
    g = torch.Generator().manual_seed(0)
    x = torch.rand(4096, 3, generator=g) * 1.2 - 0.1          # some points outside [0,1]
    x[:16] = torch.tensor([0.0, 1.0, 0.5] * 16).reshape(16, 3)
    x[16:32] = (torch.arange(16)[:, None] / 15.0).expand(16, 3)  # level-0 vertices
    rows_o, _ = tr.hashgrid_indices(x, om)
    rows_p = ops.hashgrid_rows(x.to(DEV), pm).cpu()
    assert torch.equal(rows_o, rows_p)


@pytest.mark.parametrize("hash_size,res,P,scatter,cap", [
    (16, 592, 5000, None, None),      # binned scatter (8 chunks per hashed level)
    (12, 64, 777, None, None),
    (20, 231, 4000, None, None),      # T = 2^20: partition form picked by the host (128 chunks per hashed level)
    (16, 592, 3000, "q", None),       # partition form forced on every multi-chunk level (dense ones included)
    (16, 592, 3000, "q", 64),         # ... with 64-entry queues: most contributions take the overflow fallback
    (16, 592, 3000, "a", None),       # per-corner global atomics
    (16, 592, 3000, "b", None),       # LDS bins forced for every level
    (16, 592, 5000, "r", None),       # binned + row replay: the hashed levels' corner rows stored once, replayed by the 8 chunk visits
    (14, 200, 2000, "r", None),       # ... T = 2^14: two chunks per hashed level
    (16, 592, 5000, "l", None),       # pair lists: every multi-chunk level (3 dense + 12 hashed) hashed once, {point, pair} words per chunk
    (16, 592, 3000, "l", 64),         # ... with 64-entry lists: most pairs take the overflow fallback (float atomics)
    (20, 231, 4000, "l", None),       # ... T = 2^20: up to 128 chunks per level, large dense levels (pairs that straddle chunks)
    (14, 200, 2000, "l", None),
    (16, 592, 1, "l", None),          # ... a single point, and fewer points than one pass-1 tile
    (20, 231, 63, "l", None),
])
def test_encode_forward_backward(hash_size, res, P, scatter, cap, monkeypatch):
    ops = _ops()
    form = {None: ops.SCATTER_AUTO, "q": ops.SCATTER_QUEUES, "a": ops.SCATTER_ATOMIC, "b": ops.SCATTER_BINNED,
            "r": ops.SCATTER_AUTO | ops.SCATTER_REPLAY, "l": ops.SCATTER_AUTO | ops.SCATTER_LISTS}[scatter]
    monkeypatch.setattr(ops, "SCATTER_FORM", (form, cap or 0))      # dns_encode_bwd flags / queue_cap
    om, pm = tr.grid_meta(hash_size, res), ops.GridMeta(hash_size, res)
    g = torch.Generator().manual_seed(1)
    table = (torch.rand(om.total_rows, 2, generator=g) * 2 - 1)
    x = torch.rand(P, 3, generator=g)
    gy = torch.randn(P, 80, generator=g)
    xo = x.clone().requires_grad_(True)
    to = table.clone().requires_grad_(True)
    yo = torch.cat((tr.oneblob_forward(xo, 16), tr.hashgrid_forward(xo, to, om)), -1)
    (yo * gy).sum().backward()
    xp = x.to(DEV).requires_grad_(True)
    tp = table.reshape(-1).to(DEV).requires_grad_(True)
    yp = ops.encode(xp, tp, pm, None, 16, True, True)
    (yp * gy.to(DEV)).sum().backward()
    assert_close(yp.cpu(), yo, what="encode fwd")
    assert_close(tp.grad.cpu().reshape(-1, 2), to.grad, what="d table", groups=table_level_groups(om))
    # d/dx is piecewise constant with jumps at cell boundaries; points within an ulp of a boundary may differ
    bad = ((xp.grad.cpu() - xo.grad).abs() > 1e-4 * xo.grad.abs().max()).any(-1)
    assert bad.float().mean() < 0.002, f"d x mismatch on {int(bad.sum())} points"


@pytest.mark.parametrize("n_bins", [4, 8, 16])
def test_oneblob_outside_the_unit_interval_and_small_bin_counts(n_bins):
    """OneBlob is periodic (tcnn kernel_one_blob adds the images at x -+ 1): points outside [0, 1) -- rays that leave the
    bound stay in a static-shape batch -- must encode and differentiate like the oracle.  The kernels evaluate the quartic
    kernel on its support only (five bins around x n and its images) for n_bins >= 8 and walk all edges below that; x exactly
    on bin edges and far outside (|x| up to 5: every term saturated) included."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(4000, 3, generator=g) * 3.0 - 1.0                      # [-1, 2)
    x[:64, 0] = torch.arange(64) / n_bins - 1.0                            # exactly on edges, both sides of the interval
    x[64:96] = torch.rand(32, 3, generator=g) * 10.0 - 5.0
    gy = torch.randn(4000, 3 * n_bins, generator=g)
    xo = x.clone().requires_grad_(True)
    yo = tr.oneblob_forward(xo, n_bins)
    (yo * gy).sum().backward()
    xp = x.to(DEV).requires_grad_(True)
    yp = ops.encode(xp, None, None, None, n_bins, True, False)
    (yp * gy.to(DEV)).sum().backward()
    assert_close(yp.cpu(), yo, rtol=1e-6, what=f"oneblob n={n_bins} outside [0,1)")
    # the derivative is piecewise polynomial with kinks at the kernel's edge: compare where the oracle's own value is stable
    d = (xp.grad.cpu() - xo.grad).abs()
    assert float((d > 1e-4 * xo.grad.abs().max()).float().mean()) < 0.002, f"d x mismatch on {int((d > 1e-4 * xo.grad.abs().max()).sum())} entries"


def test_oneblob_only_backward_through_the_lds_tile_at_size():
    """OneBlob-only gradients (Decoder.merge's relative points: 3 x P rows per cfg3 iteration) come in through the encoder's LDS
    tile like full rows do -- coalesced runs instead of one strided row per lane (round 5: 190 -> ~55 us for 786 432 rows of a
    [., 112] matrix; that row stride is test_merge_module_matches_oracle's).  Here: 150 001 points -- a ragged last workgroup --,
    on and outside the unit cube."""
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    P, n_bins = 150001, 16
    x = torch.rand(P, 3, generator=g) * 1.4 - 0.2
    gy = torch.randn(P, 48, generator=g)
    xo = x.clone().requires_grad_(True)
    (tr.oneblob_forward(xo, n_bins) * gy).sum().backward()
    xp = x.to(DEV).requires_grad_(True)
    yp = ops.encode(xp, None, None, None, n_bins, True, False)
    yp.backward(gy.to(DEV))
    d = (xp.grad.cpu() - xo.grad).abs()
    assert float((d > 1e-4 * xo.grad.abs().max()).float().mean()) < 0.002, f"{int((d > 1e-4 * xo.grad.abs().max()).sum())} entries differ"


def test_encode_input_gradient_from_saved_jacobian_equals_regather(monkeypatch):
    """dL/d(points) of the hash grid, two forms: the forward keeps d(features)/dx per level (ops.SAVE_DY_DX, tcnn's dy_dx,
    SURVEY K3) and the backward is a streaming dot product, or the backward gathers the 8 corners again.  Same value up to
    the order of fp32 operations (1e-5 of the scale), world-coordinate scaling included."""
    ops = _ops()
    from dns_slam_amd import synthetic
    bound = synthetic.load_bound(synthetic.ROOM0_BOUND)
    pm = ops.GridMeta(16, 592)
    g = torch.Generator().manual_seed(4)
    pts = (torch.rand(5000, 3, generator=g) * (bound[:, 1] - bound[:, 0]).float() + bound[:, 0].float()).to(DEV)
    table = (torch.rand(pm.total_rows * 2, generator=g) * 2 - 1).to(DEV)
    gy = torch.randn(5000, 80, generator=g).to(DEV)
    grads = []
    for keep in (True, False):
        monkeypatch.setattr(ops, "SAVE_DY_DX", keep)
        p_ = pts.clone().requires_grad_(True)
        t_ = table.clone().requires_grad_(True)
        (ops.encode(p_, t_, pm, bound, 16, True, True) * gy).sum().backward()
        grads.append((p_.grad.clone(), t_.grad.clone()))
    assert torch.equal(grads[0][1], grads[1][1]) or float((grads[0][1] - grads[1][1]).abs().max()) <= 1e-5 * float(grads[1][1].abs().max())
    assert float((grads[0][0] - grads[1][0]).abs().max()) <= 1e-5 * float(grads[1][0].abs().max())
    assert float(grads[0][0].abs().max()) > 0
    # grid channels only (the untiled kernel instances), normalised coordinates, a point count that is no multiple of 128
    xs = torch.rand(777, 3, generator=g).to(DEV)
    gg = torch.randn(777, 32, generator=g).to(DEV)
    gx = []
    for keep in (True, False):
        monkeypatch.setattr(ops, "SAVE_DY_DX", keep)
        x_ = xs.clone().requires_grad_(True)
        (ops.encode(x_, table, pm, None, 16, False, True) * gg).sum().backward()
        gx.append(x_.grad.clone())
    assert float((gx[0] - gx[1]).abs().max()) <= 1e-5 * float(gx[1].abs().max()) and float(gx[1].abs().max()) > 0


def test_encode_world_normalisation_fp64():
    """Fused (pts - b0)/(b1 - b0) in fp64 (slams/mapping.py:608): normalised coordinates bit-exact."""
    ops = _ops()
    from dns_slam_amd import synthetic
    bound = synthetic.load_bound(synthetic.ROOM0_BOUND)
    pm, om = ops.GridMeta(16, 592), tr.grid_meta(16, 592)
    g = torch.Generator().manual_seed(2)
    pts = torch.rand(3000, 3, generator=g) * (bound[:, 1] - bound[:, 0]).float() + bound[:, 0].float()
    table = torch.rand(om.total_rows, 2, generator=g)
    x_o = rm.normalise_points(pts, bound).float()
    rows_o, _ = tr.hashgrid_indices(x_o, om)
    y_o = torch.cat((tr.oneblob_forward(x_o, 16), tr.hashgrid_forward(x_o, table, om)), -1)
    y_p = ops.encode(pts.to(DEV), table.reshape(-1).to(DEV), pm, bound, 16, True, True)
    assert_close(y_p.cpu(), y_o, what="encode(world)")
    # the saved normalised coordinates drive the indices: check them through the rows
    from dns_slam_amd._lib import lib, ptr, stream_ptr, check
    x_p = torch.empty(3000, 3, device=DEV)
    import ctypes as C
    b6 = ops._bound6(bound)
    check(lib.dns_encode_fwd(ptr(pts.to(DEV)), b6, 3000, 16, None, None, ptr(x_p), None, 0, None, 0, None, stream_ptr()), "x")
    assert torch.equal(x_p.cpu(), x_o)
    assert torch.equal(ops.hashgrid_rows(x_p, pm).cpu(), rows_o)


@pytest.mark.parametrize("scatter", ["auto", "atomic", "queues", "lists"])
@pytest.mark.parametrize("poison", [float("nan"), float("inf")])
def test_table_gradient_propagates_non_finite(scatter, poison, monkeypatch):
    """A NaN / Inf in the upstream grid gradient must reach d_table in every scatter form: the 64-bit LDS bins of the queue form (fixed point) cannot
    carry it (fmaxf drops a NaN, the integer conversion of a non-finite product is undefined), so the transpose kernel flags
    it and the binned / queue kernels write NaN into their rows (csrc/encode.hip), as tcnn's float atomics would."""
    ops = _ops()
    pm = ops.GridMeta(16, 592)
    form = {"auto": ops.SCATTER_AUTO, "atomic": ops.SCATTER_ATOMIC, "queues": ops.SCATTER_QUEUES, "lists": ops.SCATTER_LISTS}[scatter]
    monkeypatch.setattr(ops, "SCATTER_FORM", (form, 0))
    g = torch.Generator().manual_seed(5)
    P = 3000
    x = torch.rand(P, 3, generator=g).to(DEV)
    table = (torch.rand(pm.total_rows * 2, generator=g) * 2 - 1).to(DEV).requires_grad_(True)
    gy = torch.randn(P, 32, generator=g).to(DEV)
    y = ops.encode(x, table, pm, None, 16, False, True)
    (y * gy).sum().backward()
    assert bool(torch.isfinite(table.grad).all())
    table.grad = None
    gy[1234, 17] = poison
    y = ops.encode(x, table, pm, None, 16, False, True)
    (y * gy).sum().backward()
    assert not bool(torch.isfinite(table.grad).all()), f"{scatter}: a {poison} upstream gradient left d_table finite"


@pytest.mark.parametrize("hash_size,res,P", [(16, 592, 262144), (20, 592, 100000)])
def test_pair_list_scatter_equals_the_binned_scatter_on_ray_points(hash_size, res, P, monkeypatch):
    """The pair-list form (DNS_SCATTER_LISTS) against the LDS-bin sweep on a step-sized batch of RAY samples (clustered on the dense
    levels: their lists fill unevenly and pairs straddle chunk boundaries) with a tenth of the gradients exactly zero (those points
    enter no list): both forms sum the same fp32 products in float64 bins, so every table entry agrees to the last-place noise of
    the final float atomics (1e-6 of the level's largest entry)."""
    ops = _ops()
    pm = ops.GridMeta(hash_size, res)
    g = torch.Generator().manual_seed(11)
    o = torch.rand(P // 64, 1, 3, generator=g) * 0.3 + 0.35
    d = torch.randn(P // 64, 1, 3, generator=g) * 0.3
    t = torch.linspace(0, 1, 64)[None, :, None]
    x = (o + d * t).reshape(-1, 3).clamp(0, 1).to(DEV)
    gy = torch.randn(x.shape[0], 32, generator=g)
    gy[torch.rand(x.shape[0], generator=g) < 0.1] = 0.0
    gy = gy.to(DEV)
    grads = []
    for form in (ops.SCATTER_BINNED, ops.SCATTER_LISTS):
        monkeypatch.setattr(ops, "SCATTER_FORM", (form, 0))
        table = torch.rand(pm.total_rows * 2, generator=torch.Generator().manual_seed(1)).to(DEV).requires_grad_(True)
        y = ops.encode(x, table, pm, None, 16, False, True)
        y.backward(gy)
        grads.append(table.grad.reshape(-1, 2).clone())
    off = 0
    for l in range(pm.c.n_levels):
        n = pm.c.size[l]
        a, b = grads[0][off:off + n], grads[1][off:off + n]
        assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max()), f"level {l}: {float((a - b).abs().max())} vs max {float(a.abs().max())}"
        off += n


def test_encode_known_answers():
    ops = _ops()
    pm = ops.GridMeta(16, 592)
    x = torch.rand(100, 3, device=DEV)
    zero = ops.encode(x, torch.zeros(pm.total_rows * 2, device=DEV), pm, None, 16, False, True)
    assert torch.count_nonzero(zero) == 0
    one = ops.encode(x, torch.ones(pm.total_rows * 2, device=DEV), pm, None, 16, False, True)
    assert torch.allclose(one, torch.ones_like(one), atol=1e-5)
    pe = ops.encode(x, None, None, None, 16, True, False).reshape(100, 3, 16)
    assert torch.allclose(pe.sum(-1), torch.ones(100, 3, device=DEV), atol=1e-5)
    assert ops.encode(torch.zeros(0, 3, device=DEV), torch.zeros(pm.total_rows * 2, device=DEV), pm).shape == (0, 80)


# ----------------------------------------------------------------------------------------- MLP
SHAPES = [(80, 33, 32, 1), (112, 3, 32, 1), (112, 40, 32, 1), (112, 8, 32, 1), (80, 33, 64, 2), (112, 3, 64, 2),
          (112, 8, 64, 2), (112, 32, 32, 1), (80, 33, 32, 2), (112, 40, 64, 1), (16, 64, 32, 1)]


@pytest.mark.parametrize("n_in,n_out,nn,nl", SHAPES)
@pytest.mark.parametrize("P,save_hidden", [(1000, True), (129, True), (777, False)])
def test_mlp_forward_backward(n_in, n_out, nn, nl, P, save_hidden, monkeypatch):
    ops = _ops()
    monkeypatch.setattr(ops, "MLP_SAVE_HIDDEN", save_hidden)     # both backward variants: saved activations / recompute
    g = torch.Generator().manual_seed(n_in * 7 + n_out * 3 + nn + nl)
    params = tr.mlp_init(n_in, n_out, nn, nl, g)
    assert params.numel() == ops.mlp_param_count(n_in, n_out, nn, nl)
    x = torch.randn(P, n_in, generator=g)
    gy = torch.randn(P, n_out, generator=g)
    xo, po = x.clone().requires_grad_(True), params.clone().requires_grad_(True)
    yo = tr.mlp_forward(xo, po, n_in, n_out, nn, nl)
    (yo * gy).sum().backward()
    xp, pp = x.to(DEV).requires_grad_(True), params.to(DEV).requires_grad_(True)
    yp = ops.mlp(xp, pp, n_in, n_out, nn, nl)
    (yp * gy.to(DEV)).sum().backward()
    assert_close(yp.cpu(), yo, what="mlp fwd")
    assert_close(xp.grad.cpu(), xo.grad, what="mlp dx")
    op = tr.mlp_out_padded(n_out)
    used = slice(0, params.numel() - (op - n_out) * nn)            # padded output rows get no gradient
    assert_close(pp.grad.cpu()[used], po.grad[used], what="mlp dparams")
    assert torch.count_nonzero(pp.grad.cpu()[used.stop:]) == 0


def test_mlp_strided_input_view():
    """x may be a column view of a wider buffer (the [P,80] encode buffer inside a [P,112] one)."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    params = tr.mlp_init(80, 33, 32, 1, g)
    big = torch.randn(500, 112, generator=g)
    yo = tr.mlp_forward(big[:, :80], params, 80, 33, 32, 1)
    yp = ops.mlp(big.to(DEV)[:, :80], params.to(DEV), 80, 33, 32, 1)
    assert_close(yp.cpu(), yo, what="mlp strided")


@pytest.mark.parametrize("nn,nl", [(32, 1), (64, 2)])
def test_mlp_grouped_matches_per_class_loop(nn, nl):
    """Per-class fine decoders (slams/mapping.py:590-601): routing, the >1-point rule, zeros elsewhere."""
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    G, P = 5, 3000
    pool = torch.stack([tr.mlp_init(80, 33, nn, nl, g) for _ in range(G)])
    x = torch.randn(P, 80, generator=g)
    slot = torch.randint(0, G - 1, (P,), generator=g)
    slot[7] = G - 1                      # a class with exactly one point -> zeros (index.sum() > 1 rule)
    slot[11] = -1                        # no decoder
    gy = torch.randn(P, 33, generator=g)
    xo, po = x.clone().requires_grad_(True), pool.clone().requires_grad_(True)
    yo = torch.zeros(P, 33)
    for c in range(G):
        idx = torch.nonzero(slot == c).reshape(-1)
        if idx.numel() > 1:
            yo = yo.index_put((idx,), tr.mlp_forward(xo[idx], po[c], 80, 33, nn, nl))
    (yo * gy).sum().backward()
    xp, pp = x.to(DEV).requires_grad_(True), pool.to(DEV).requires_grad_(True)
    yp = ops.mlp_grouped(xp, pp, slot.to(DEV), 80, 33, nn, nl)
    (yp * gy.to(DEV)).sum().backward()
    assert_close(yp.cpu(), yo, what="grouped fwd")
    assert torch.count_nonzero(yp[7]) == 0 and torch.count_nonzero(yp[11]) == 0
    assert_close(xp.grad.cpu(), xo.grad, what="grouped dx")
    used = 80 * nn + (nl - 1) * nn * nn + 33 * nn
    assert_close(pp.grad.cpu()[:, :used], po.grad[:, :used], what="grouped dparams")


def _mlp_fp16_emulation(x, w, gy, n_in, n_out, nn, nl):
    """What the fp16 mode computes, restated with torch: every matrix-product OPERAND rounded to fp16 (weights, layer
    inputs, output / hidden gradients), products and sums exact (float64), ReLU masks from the fp32-stored activations;
    weight gradients from the fp32 dH and fp32 activations / inputs."""
    q = lambda t: t.to(torch.float16).to(torch.float64)
    f32 = lambda t: t.to(torch.float32)
    o = 0
    Ws = []
    for r, c in [(nn, n_in)] + [(nn, nn)] * (nl - 1) + [(n_out, nn)]:
        Ws.append(w[o:o + r * c].reshape(r, c))
        o += r * c
    acts = [x]
    h = x
    for W in Ws[:-1]:
        h = f32(torch.relu(q(h) @ q(W).T))
        acts.append(h)
    y = f32(q(h) @ q(Ws[-1]).T)
    d = gy
    dWs = [None] * len(Ws)
    for li in range(len(Ws) - 1, -1, -1):
        dWs[li] = f32(d.double().T @ acts[li].double())
        d_in = f32(q(d) @ q(Ws[li]))
        if li > 0:
            d_in = d_in * (acts[li] > 0).float()
        d = d_in
    return y, d, torch.cat([t.reshape(-1) for t in dWs])


@pytest.mark.parametrize("n_in,n_out,nn,nl", [(80, 33, 64, 2), (112, 8, 64, 2), (112, 40, 32, 1), (80, 3, 64, 1), (16, 64, 32, 2)])
def test_mlp_fp16_mode(n_in, n_out, nn, nl):
    """BASELINE configs[4] / SURVEY D11: the fp16-operand MFMA mode (tcnn's own precision: fp16 operands, fp32
    accumulate).  (i) Against a torch restatement that rounds the same operands to fp16: 2e-3 of the scale (summation
    order; a hidden unit at ~0 may still fall on the other side of the ReLU: <= 0.5 % of the points may be off).
    (ii) Against the exact-fp32 kernels: within 1e-2 (outputs) / 5e-2 rms (gradients: ReLU flips).  (iii) Output
    gradients of 1e-7 -- far below fp16's normal range -- give the same input gradient up to scale: the per-tile
    power-of-two scaling of the backward."""
    ops = _ops()
    g = torch.Generator().manual_seed(13)
    P = 2000
    w = tr.mlp_init(n_in, n_out, nn, nl, g).to(DEV)
    x = torch.randn(P, n_in, generator=g).to(DEV)
    gy = torch.randn(P, n_out, generator=g).to(DEV)

    def run(fp16, scale=1.0):
        xp, wp = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = ops.mlp(xp, wp, n_in, n_out, nn, nl, fp16=fp16)
        (y * (gy * scale)).sum().backward()
        used = n_in * nn + (nl - 1) * nn * nn + n_out * nn
        return y.detach(), xp.grad, wp.grad[:used]

    r16, r32 = run(True), run(False)
    emu = _mlp_fp16_emulation(x, w, gy, n_in, n_out, nn, nl)
    for a, b, name in zip(r16, emu, ("y", "dx", "dparams")):
        scale = float(b.abs().max())
        d = (a - b).abs()
        if name == "dx":
            bad = float((d.max(1)[0] > 2e-3 * scale).float().mean())
            assert bad <= 0.005, f"fp16 mode vs fp16 emulation, dx: {bad * 100:.2f} % of the points off by > 2e-3"
        else:
            rms = float(d.pow(2).mean().sqrt()) / float(b.pow(2).mean().sqrt())
            assert rms <= 2e-3, f"fp16 mode vs fp16 emulation, {name}: relative rms error {rms:.3e}"
    assert float((r16[0] - r32[0]).abs().max()) <= 1e-2 * float(r32[0].abs().max())
    assert float((r16[0] - r32[0]).abs().max()) > 0, "fp16 mode returned bit-identical outputs: the fp32 kernel ran instead"
    for a, b in zip(r16[1:], r32[1:]):
        assert float((a - b).pow(2).mean().sqrt()) <= 5e-2 * float(b.pow(2).mean().sqrt())
    tiny = run(True, 1e-7)
    d = (tiny[1] / 1e-7 - r16[1]).abs()
    assert float((d.max(1)[0] > 5e-3 * float(r16[1].abs().max())).float().mean()) <= 0.01, "tiny output gradients lost in fp16"
    den = run(True, 1e-41)                       # denormal output gradients: the power-of-two scale must stay finite
    assert all(bool(torch.isfinite(t).all()) for t in den), "denormal-small output gradients produced inf / nan"


@pytest.mark.parametrize("nn,nl,P,fine_used", [(64, 2, 1500, True), (32, 1, 333, True), (64, 2, 700, False)])
def test_render_nets_two_segment_input_and_in_place_gradient_sums(nn, nl, P, fine_used):
    """ops.render_nets vs the oracle MLPs wired like Mapper.renderer (slams/mapping.py:616-627, models/decoder.py:
    123-124): coarse(buf), per-class fine(buf), colour / logit(cat(pe, fine[:, 1:], pixel)), raw = cat(sigmoid(colour),
    fine[:, 0:1]).  Covers the two-segment input of dns_mlp_fwd/bwd (48 | 64 columns), accumulate_dx bits 0 and 1, the
    grouped network adding into the same input gradient, the pixel-feature gradient, the colour network writing into
    the [P, 4] compositing rows (ldy 4) and -- fine_used False: no loss on the latents, as in the tracker -- the fine
    network's output gradient read as a strided view (lddy 68) of the feature-gradient buffer."""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    G, pe_dim, hid, C, n_class = 3, 48, 32, 32, 8
    shp_c, shp_f = (80, hid + 1, nn, nl), (80, hid + 1, nn, nl)
    shp_col, shp_log = (pe_dim + hid + C, 3, nn, nl), (pe_dim + hid + C, n_class, nn, nl)
    cp = tr.mlp_init(*shp_c, g)
    pool = torch.stack([tr.mlp_init(*shp_f, g) for _ in range(G)])
    colp, logp = tr.mlp_init(*shp_col, g), tr.mlp_init(*shp_log, g)
    buf, pix = torch.randn(P, 80, generator=g), torch.randn(P, C, generator=g)
    slot = torch.randint(0, G, (P,), generator=g)
    slot[5] = -1
    gw = [torch.randn(P, n, generator=g) for n in (hid + 1, hid + 1, 4, n_class)]
    if not fine_used:
        gw[1] = None
    total = lambda outs, dev: sum((o * w.to(dev)).sum() for o, w in zip(outs, gw) if w is not None)

    # oracle wiring
    bo, xo = buf.clone().requires_grad_(True), pix.clone().requires_grad_(True)
    po = [t.clone().requires_grad_(True) for t in (cp, pool, colp, logp)]
    coarse = tr.mlp_forward(bo, po[0], *shp_c)
    fine = torch.zeros(P, hid + 1)
    for c in range(G):
        idx = torch.nonzero(slot == c).reshape(-1)
        if idx.numel() > 1:
            fine = fine.index_put((idx,), tr.mlp_forward(bo[idx], po[1][c], *shp_f))
    xin = torch.cat((bo[:, :pe_dim], fine[:, 1:], xo), -1)
    raw = torch.cat((torch.sigmoid(tr.mlp_forward(xin, po[2], *shp_col)), fine[:, 0:1]), -1)
    outs_o = [coarse, fine, raw, tr.mlp_forward(xin, po[3], *shp_log)]
    total(outs_o, "cpu").backward()

    bp, xp = buf.to(DEV).requires_grad_(True), pix.to(DEV).requires_grad_(True)
    pp = [t.to(DEV).requires_grad_(True) for t in (cp, pool, colp, logp)]
    outs_p = ops.render_nets(bp, xp, pp[0], pp[1], pp[2], pp[3], slot.to(DEV), pe_dim, shp_c, shp_f, shp_col, shp_log)
    total(outs_p, DEV).backward()
    for a, b, name in zip(outs_p, outs_o, ("coarse", "fine", "raw", "logit")):
        assert_close(a.cpu(), b, what=f"render_nets {name}")
    assert_close(bp.grad.cpu(), bo.grad, what="render_nets d_buf")
    assert_close(xp.grad.cpu(), xo.grad, what="render_nets d_pixel")
    for a, b, shp, name in zip(pp, po, (shp_c, shp_f, shp_col, shp_log), ("coarse", "fine pool", "colour", "logit")):
        used = shp[0] * nn + (nl - 1) * nn * nn + shp[1] * nn
        assert_close(a.grad.cpu()[..., :used], b.grad[..., :used], what=f"render_nets d_params {name}")


# ----------------------------------------------------------------------------------------- compositing
def test_composite_golden(golden_dir):
    """raw2nerf_color outputs and input gradient of the IMPORTED reference."""
    ops = _ops()
    gd = np.load(os.path.join(golden_dir, "raw2nerf_color.npz"))
    for ci in range(int(gd["n_cases"])):
        p = f"c{ci}_"
        raw = _t(gd[p + "raw"]).to(DEV).requires_grad_(True)
        z = _t(gd[p + "z"]).to(DEV)
        depth, var, rgb, w, _ = ops.composite(raw, z, None)
        for a, k in ((depth, "depth"), (var, "var"), (rgb, "rgb"), (w, "weights")):
            assert_close(a.cpu(), _t(gd[p + k]), what=f"case {ci} {k}")
        loss = (depth * _t(gd[p + "g_depth"]).to(DEV)).sum() + (var * _t(gd[p + "g_var"]).to(DEV)).sum() \
            + (rgb * _t(gd[p + "g_rgb"]).to(DEV)).sum() + (w * _t(gd[p + "g_w"]).to(DEV)).sum()
        loss.backward()
        assert_close(raw.grad.cpu(), _t(gd[p + "grad_raw"]), what=f"case {ci} grad_raw")


@pytest.mark.parametrize("N,S,C", [(300, 47, 40), (64, 64, 8), (33, 128, 8), (5, 200, 3), (7, 1, 4), (2000, 64, 0)])
def test_composite_vs_oracle(N, S, C):
    ops = _ops()
    g = torch.Generator().manual_seed(N + S + C)
    raw = torch.randn(N, S, 4, generator=g)
    raw[..., 3] *= 0.3
    z = torch.sort(torch.rand(N, S, generator=g) * 4 + 0.1, -1)[0]
    logits = torch.randn(N, S, C, generator=g) if C else None
    gs = [torch.randn(N, generator=g), torch.randn(N, generator=g), torch.randn(N, 3, generator=g), torch.randn(N, max(C, 1), generator=g)]
    ro = raw.clone().requires_grad_(True)
    lo = logits.clone().requires_grad_(True) if C else None
    d, v, c, w = rm.raw2nerf_color(ro, z)
    loss = (d * gs[0]).sum() + (v * gs[1]).sum() + (c * gs[2]).sum()
    if C:
        sem_o = torch.sum(w[..., None] * lo, -2)
        loss = loss + (sem_o * gs[3]).sum()
    loss.backward()
    rp = raw.to(DEV).requires_grad_(True)
    lp = logits.to(DEV).requires_grad_(True) if C else None
    dp, vp, cp, wp, sp = ops.composite(rp, z.to(DEV), lp)
    lossp = (dp * gs[0].to(DEV)).sum() + (vp * gs[1].to(DEV)).sum() + (cp * gs[2].to(DEV)).sum()
    if C:
        lossp = lossp + (sp * gs[3].to(DEV)).sum()
    lossp.backward()
    assert_close(dp.cpu(), d, what="depth")
    assert_close(vp.cpu(), v, what="var")
    assert_close(cp.cpu(), c, what="rgb")
    assert_close(wp.cpu(), w, what="weights")
    assert abs(float(wp.sum(-1).mean()) - 1.0) < 1e-5
    assert_close(rp.grad.cpu(), ro.grad, what="d raw")
    if C:
        assert_close(sp.cpu(), sem_o, what="sem")
        assert_close(lp.grad.cpu(), lo.grad, what="d logits")


def test_composite_known_answers_and_nan():
    ops = _ops()
    raw = torch.zeros(3, 8, 4, device=DEV)              # alpha = 0.5 everywhere -> w_i ~ 2^-i (1e-10 aside)
    z = torch.arange(8, device=DEV).float().expand(3, 8).contiguous()
    _, _, _, w, _ = ops.composite(raw, z, None)
    ref = 0.5 ** torch.arange(1, 9).float()
    assert torch.allclose(w[0].cpu(), ref / ref.sum(), atol=1e-6)
    one = ops.composite(torch.randn(4, 1, 4, device=DEV), torch.rand(4, 1, device=DEV), None)[3]
    assert torch.allclose(one, torch.ones_like(one))      # single-sample ray -> weight 1
    dead = torch.zeros(2, 8, 4, device=DEV)
    dead[..., 3] = -20.0                                  # every alpha underflows: 0/0, reference gives NaN (D9)
    d = ops.composite(dead, z[:2].contiguous(), None)[0]
    d_ref = rm.raw2nerf_color(dead.cpu(), z[:2].cpu())[0]
    assert torch.isnan(d_ref).all() and torch.isnan(d).all()


# ----------------------------------------------------------------------------------------- ray generation + sampling
def _scene_small(seed, H=48, W=64, K=3):
    g = torch.Generator().manual_seed(seed)
    color = torch.rand(K, H, W, 3, generator=g)
    depth = torch.rand(K, H, W, generator=g) * 4 + 0.3
    depth[torch.rand(K, H, W, generator=g) < 0.05] = 0.0
    label = torch.randint(0, 6, (K, H, W), generator=g).float()
    q = torch.randn(K, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True) * (1 + 0.1 * torch.rand(K, 1, generator=g))     # not exactly unit
    T = torch.randn(K, 3, generator=g) * 0.5
    return color, depth, label, q, T


@pytest.mark.parametrize("nu,ns", [(32, 15), (48, 16), (96, 32), (0, 15), (22, 10)])
def test_raygen_sample_bit_exact_vs_oracle(nu, ns):
    ops = _ops()
    K, H, W, npf = 3, 48, 64, 200
    color, depth, label, q, T = _scene_small(nu + ns)
    bound = torch.tensor([[-3.0, 3.5], [-2.5, 4.0], [-2.0, 2.2]], dtype=torch.float64)
    cam = (50.0, 52.0, (W - 1) / 2.0, (H - 1) / 2.0)
    g = torch.Generator().manual_seed(9)
    idx = torch.randint(H * W, (K * npf,), generator=g)
    t = torch.rand(ns, generator=g)
    t[ns // 2 + 1] = 0.5
    t0 = torch.rand(ns, generator=g)
    tu = torch.linspace(0.0, 1.0, steps=nu) if nu else None
    outs = ops.raygen_sample(q.to(DEV), T.to(DEV), idx.to(DEV), color.to(DEV), depth.to(DEV), label.to(DEV), cam, bound,
                             (0, H, 0, W), npf, tu.to(DEV) if nu else None, t.to(DEV), t0.to(DEV))
    rays_o, rays_d, pts, gt_color, gt_depth, gt_label, inside, z = [o.cpu() for o in outs]
    for f in range(K):
        sl = slice(f * npf, (f + 1) * npf)
        img5 = torch.cat((color[f], depth[f][..., None], label[f][..., None]), -1)
        R = rm.rotation_from_quad(q[f])
        i, j = rm.uv_from_indices(idx[sl], 0, H, 0, W)
        px = rm.gather_pixels(idx[sl], img5, 0, H, 0, W)
        ro, rd = rm.rays_from_uv(i, j, R, T[f], *cam)
        far, ins = rm.box_far(ro, rd, px[:, 3], bound)
        zo = rm.sample_along_rays(px[:, 3], nu, ns, far, t, t0)
        assert torch.equal(gt_color[sl], px[:, :3]) and torch.equal(gt_depth[sl], px[:, 3])
        assert torch.equal(gt_label[sl], px[:, 4].long())
        assert_close(rays_d[sl], rd, rtol=1e-6, what="rays_d")       # quat->R summation order may differ by an ulp
        assert torch.equal(rays_o[sl], ro.contiguous())
        assert torch.equal(inside[sl].bool(), ins)
        if torch.equal(rays_d[sl], rd):
            assert torch.equal(z[sl], zo), f"frame {f}: z not bit-exact"
            assert torch.equal(pts[sl], rm.points_from_rays(ro, rd, zo))
        else:
            assert_close(z[sl], zo, rtol=1e-6, what="z")
        assert (z[sl][:, 1:] >= z[sl][:, :-1]).all()                  # sorted ascending


def test_sample_along_rays_golden_bit_exact(golden_dir):
    """sample_along_rays of the IMPORTED reference (15 cases: zero depths, negative / huge far_bb, S up to 128)."""
    from dns_slam_amd import common
    gd = np.load(os.path.join(golden_dir, "sample_along_rays.npz"))
    for ci in range(int(gd["n_cases"])):
        p = f"c{ci}_"
        nu, ns = [int(v) for v in gd[p + "n"]]
        t = _t(gd[p + "t_raw"]).clone()
        if not torch.any(t == 0.5):
            t[ns // 2 + 1] = 0.5
        z = common.sample_along_rays(_t(gd[p + "depth"]).to(DEV), nu, ns, _t(gd[p + "far_bb"]).to(DEV), DEV,
                                     jitter=(t, _t(gd[p + "t_zero"])))
        assert torch.equal(z.cpu(), _t(gd[p + "z"])), f"golden case {ci} ({nu}+{ns})"


def test_raygen_pose_gradient():
    ops = _ops()
    K, H, W, npf, nu, ns = 2, 48, 64, 150, 8, 5
    color, depth, label, q, T = _scene_small(77, K=K)
    bound = torch.tensor([[-3.0, 3.5], [-2.5, 4.0], [-2.0, 2.2]], dtype=torch.float64)
    cam = (50.0, 52.0, (W - 1) / 2.0, (H - 1) / 2.0)
    g = torch.Generator().manual_seed(10)
    idx = torch.randint(H * W, (K * npf,), generator=g)
    t, t0 = torch.rand(ns, generator=g), torch.rand(ns, generator=g)
    tu = torch.linspace(0.0, 1.0, steps=nu)
    gp = torch.randn(K * npf, nu + ns, 3, generator=g)
    gd_, go = torch.randn(K * npf, 3, generator=g), torch.randn(K * npf, 3, generator=g)
    qp, Tp = q.to(DEV).requires_grad_(True), T.to(DEV).requires_grad_(True)
    outs = ops.raygen_sample(qp, Tp, idx.to(DEV), color.to(DEV), depth.to(DEV), label.to(DEV), cam, bound, (0, H, 0, W),
                             npf, tu.to(DEV), t.to(DEV), t0.to(DEV))
    ((outs[2] * gp.to(DEV)).sum() + (outs[1] * gd_.to(DEV)).sum() + (outs[0] * go.to(DEV)).sum()).backward()
    z = outs[7].cpu()
    qo, To = q.clone().requires_grad_(True), T.clone().requires_grad_(True)
    loss = 0
    for f in range(K):
        sl = slice(f * npf, (f + 1) * npf)
        i, j = rm.uv_from_indices(idx[sl], 0, H, 0, W)
        ro, rd = rm.rays_from_uv(i, j, rm.rotation_from_quad(qo[f]), To[f], *cam)
        loss = loss + (rm.points_from_rays(ro, rd, z[sl]) * gp[sl]).sum() + (rd * gd_[sl]).sum() + (ro * go[sl]).sum()
    loss.backward()
    assert_close(qp.grad.cpu(), qo.grad, what="d quat")
    assert_close(Tp.grad.cpu(), To.grad, what="d T")


# ----------------------------------------------------------------------------------------- fused Adam / TV / grouping
def test_fused_adam_matches_torch_adam():
    """csrc/adam.hip vs torch.optim.Adam (reference slams/mapping.py:464): 25 steps, three lr groups, a frozen tensor."""
    from dns_slam_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(1706624,), (64, 80), (33, 64), (4,), (3,)]
    p_ref = [torch.randn(*s, generator=g).to(DEV).requires_grad_(True) for s in shapes]
    p_fus = [p.detach().clone().requires_grad_(True) for p in p_ref]
    frozen_r, frozen_f = torch.ones(5, device=DEV), torch.ones(5, device=DEV)
    groups = lambda ps, fr: [{"params": ps[:3], "lr": 5e-3}, {"params": [ps[3], fr], "lr": 5e-4}, {"params": [ps[4]], "lr": 1e-3}]
    o_ref = torch.optim.Adam(groups(p_ref, frozen_r))
    o_fus = FusedAdam(groups(p_fus, frozen_f))
    for it in range(25):
        for a, b in zip(p_ref, p_fus):
            gr = torch.randn(a.shape, generator=g).to(DEV) * (10.0 ** ((it % 5) - 3))
            a.grad, b.grad = gr.clone(), gr.clone()
        o_ref.step()
        o_fus.step()
    for a, b in zip(p_ref, p_fus):
        assert_close(b.detach().cpu(), a.detach().cpu(), rtol=1e-6, what="FusedAdam parameter")
    assert torch.equal(frozen_f, torch.ones(5, device=DEV))


@pytest.mark.parametrize("n,ld", [(11, 33), (63, 33), (5, 1)])
def test_tv_smoothness_matches_torch(n, ld):
    ops = _ops()
    g = torch.Generator().manual_seed(n)
    lat = torch.randn(n ** 3, ld, generator=g).to(DEV).requires_grad_(True)
    out = ops.tv_smoothness(lat, n, n + 1)
    out.backward(torch.tensor(0.7, device=DEV))
    lt = lat.detach().clone().requires_grad_(True)
    occ = lt[:, 0:1].reshape(n, n, n, 1)
    ref = (torch.pow(occ[1:] - occ[:-1], 2).sum() + torch.pow(occ[:, 1:] - occ[:, :-1], 2).sum()
           + torch.pow(occ[:, :, 1:] - occ[:, :, :-1], 2).sum()) / ((n + 1) ** 3)
    ref.backward(torch.tensor(0.7, device=DEV))
    assert abs(float(out) - float(ref)) <= 1e-5 * abs(float(ref))
    assert_close(lat.grad.cpu(), lt.grad.cpu(), rtol=1e-5, what="TV gradient")


def test_group_slots_layout():
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    P, G = 20000, 7
    slot = torch.randint(-1, G - 1, (P,), generator=g)
    slot[123] = G - 1                                       # a group with a single point -> skipped (min_count 2)
    ri, tg, n_slots = ops.group_slots(slot.to(DEV), G, 2)
    ri, tg = ri.cpu().long(), tg.cpu().long()
    assert n_slots % 128 == 0 and ri.numel() == n_slots and tg.numel() == n_slots // 128
    placed = ri[ri >= 0]
    assert torch.equal(torch.sort(placed)[0], torch.nonzero(slot >= 0).reshape(-1))      # every routed point exactly once
    for t in range(n_slots // 128):
        rows = ri[t * 128:(t + 1) * 128]
        rows = rows[rows >= 0]
        if rows.numel():
            grp = torch.unique(slot[rows])
            assert grp.numel() == 1                          # one weight set per 128-slot tile
            assert int(tg[t]) == (int(grp) if int((slot == int(grp)).sum()) >= 2 else -1)
        else:
            assert int(tg[t]) == -1 or int((slot == int(tg[t])).sum()) >= 2


def test_group_scatter_stale_cursor_is_an_error_code_not_an_out_of_bounds_store():
    """A consumer of DEVICE counters must not trust them: the scatter step of dns_group_slots driven with a deliberately stale
    cursor (its group would run 990 slots past the table) stores nothing out of bounds, sets the sticky device error word,
    and from then on every entry point returns DNS_E_LAUNCH until the word is cleared (include/dns_hip.h, dns_device_error)."""
    import ctypes as C
    from dns_slam_amd._lib import check, ensure_init, lib, ptr, stream_ptr
    ensure_init()
    assert lib.dns_device_error(1) == 0
    P, G, n_slots, guard = 1000, 3, 1280, 4096
    slot = torch.zeros(P, dtype=torch.int64, device=DEV)                 # every point in group 0
    table = torch.full((n_slots + guard,), -7, dtype=torch.int32, device=DEV)
    cursor = torch.tensor([n_slots - 10, 0, 0], dtype=torch.int32, device=DEV)     # stale: room for 10 of the 1000 points
    rc = lib.dns_group_scatter(ptr(slot), P, G, ptr(cursor), n_slots, ptr(table), stream_ptr())
    torch.cuda.synchronize()
    assert rc in (0, -2)
    t = table.cpu()
    assert int((t[:n_slots - 10] != -7).sum()) == 0 and int((t[n_slots:] != -7).sum()) == 0     # nothing outside [n_slots-10, n_slots)
    assert int((t[n_slots - 10:n_slots] >= 0).sum()) == 10                                       # the 10 that fit were placed
    assert lib.dns_device_error(0) & 1
    # sticky at the C ABI: an unrelated, well-formed call now reports the fault (and keeps reporting it) ...
    ops = _ops()
    small = torch.zeros(4, 4, device=DEV)
    assert lib.dns_rgb_sigmoid(ptr(small), 4, stream_ptr()) == -2 and lib.dns_rgb_sigmoid(ptr(small), 4, stream_ptr()) == -2
    assert lib.dns_device_error(0) & 1
    # ... the Python wrapper turns it into ONE exception that names the word and clears it (round 4, _lib.check)
    with pytest.raises(RuntimeError, match="device error word 0x1"):
        ops.group_slots(torch.zeros(256, dtype=torch.int64, device=DEV), 2, 2)
    assert lib.dns_device_error(0) == 0
    ri, tg, ns = ops.group_slots(torch.zeros(256, dtype=torch.int64, device=DEV), 2, 2)
    torch.cuda.synchronize()
    assert int((ri >= 0).sum()) == 256 and lib.dns_device_error(0) == 0


# ----------------------------------------------------------------------------------------- degenerate sizes, argument errors
def test_empty_batches_and_argument_errors():
    """Zero points / rays / slots pass through every op (empty outputs, zero parameter gradients, no launch fault), a
    batch whose points all miss a fine decoder gives zeros, and malformed arguments come back as an error code with a
    message from dns_last_error -- nothing is launched on bad shapes (the reference's tcnn modules raise likewise)."""
    ops = _ops()
    from dns_slam_amd import _lib
    lib = _lib.lib
    m = ops.GridMeta(12, 64)
    table = torch.rand(m.total_rows * 2, device=DEV, requires_grad=True)
    # P = 0
    y = ops.encode(torch.empty(0, 3, device=DEV), table, m, None, 16, True, True)
    assert y.shape == (0, 48 + 2 * m.n_levels)
    y.sum().backward()
    assert table.grad is not None and float(table.grad.abs().max()) == 0.0
    w = (torch.randn(ops.mlp_param_count(80, 33, 64, 2), device=DEV) * 0.1).requires_grad_(True)
    out = ops.mlp(torch.empty(0, 80, device=DEV), w, 80, 33, 64, 2)
    assert out.shape == (0, 33)
    out.sum().backward()
    assert float(w.grad.abs().max()) == 0.0
    # one point (a single ragged tile), and a grouped call where no point has a network
    x1 = torch.randn(1, 80, device=DEV)
    assert_close(ops.mlp(x1, w.detach(), 80, 33, 64, 2).cpu(), tr.mlp_forward(x1.cpu(), w.detach().cpu(), 80, 33, 64, 2),
                 what="one-point MLP")
    pool = torch.randn(3, ops.mlp_param_count(80, 33, 64, 2), device=DEV) * 0.1
    none = ops.mlp_grouped(torch.randn(200, 80, device=DEV), pool, torch.full((200,), -1, device=DEV, dtype=torch.int64),
                           80, 33, 64, 2)
    assert none.shape == (200, 33) and float(none.abs().max()) == 0.0
    # N = 0 rays
    d, v, rgb, wts, sem = ops.composite(torch.empty(0, 64, 4, device=DEV), torch.empty(0, 64, device=DEV),
                                        torch.empty(0, 64, 8, device=DEV))
    assert d.shape == (0,) and rgb.shape == (0, 3) and wts.shape == (0, 64) and sem.shape == (0, 8)
    # argument errors: unsupported width, misaligned two-segment input, a level table that does not fit
    x = torch.randn(256, 80, device=DEV)
    yb = torch.empty(256, 33, device=DEV)
    p = lambda t: _lib.ptr(t)
    rc = lib.dns_mlp_fwd(p(x), 80, None, 0, 0, p(w), 80, 33, 48, 2, p(yb), 33, 256, None, None, 0, None, 0, None)
    assert rc != 0 and b"unsupported shape" in lib.dns_last_error()
    # (the FORWARD takes a second segment of any 4-byte alignment -- a column slice of another matrix --, but not one narrower than
    #  the columns it must supply; the backward, whose d_x2 stores are 16 bytes wide, keeps the 16-byte rule)
    rc = lib.dns_mlp_fwd(p(x), 80, p(x[:, 1:]), 16, 48, p(w), 80, 33, 64, 2, p(yb), 33, 256, None, None, 0, None, 0, None)
    assert rc != 0 and b"x2 must be 4-byte aligned with ldx2 >= n_in - n_in1" in lib.dns_last_error()
    dyb, dxb, dx2b, wsb = torch.zeros(256, 33, device=DEV), torch.empty(256, 80, device=DEV), torch.empty(256, 80, device=DEV), torch.empty(256 * 64, device=DEV)
    rc = lib.dns_mlp_bwd(p(x), 80, p(x[:, 1:]), 80, 48, p(dyb), 33, p(w), 80, 33, 64, 2, p(dxb), 80, p(dx2b), 80, None, p(wsb), 256, None, None,
                         0, None, 0, None)
    assert rc != 0 and b"x2 must be 16-byte aligned" in lib.dns_last_error()
    rc = lib.dns_mlp_fwd(p(x), 80, None, 0, 0, p(w), 80, 33, 64, 2, p(yb), 16, 256, None, None, 0, None, 0, None)
    assert rc != 0 and b"ldy" in lib.dns_last_error()
    torch.cuda.synchronize()                         # nothing faulted


def test_kernel_timing_spans():
    """dns_kernel_timing: every kernel of an armed call is bracketed on its launch stream; names and positive durations come
    back per launch, attributed to the entry point that launched them (bench.py's per-kernel roofline)."""
    ops = _ops()
    pm = ops.GridMeta(16, 592)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(4096, 3, generator=g).to(DEV)
    table = torch.rand(pm.total_rows * 2, generator=g).to(DEV).requires_grad_(True)
    ops.timer.arm(kernels=True)
    y = ops.encode(x, table, pm, None, 16, True, True)
    y.sum().backward()
    torch.cuda.synchronize()
    per_entry = ops.timer.disarm()
    spans = ops.timer.kernel_spans
    assert set(per_entry) == {"dns_encode_fwd", "dns_encode_bwd"}
    names = [(e, k.split("<")[0]) for e, k, ms, units, info in spans]
    assert ("dns_encode_fwd", "encode_fwd_kernel") in names
    assert ("dns_encode_bwd", "dgrid_transpose_kernel") in names and ("dns_encode_bwd", "hashgrid_bwd_binned_kernel") in names
    assert all(ms > 0 for _, _, ms, _, _ in spans) and all(units == 4096 for _, _, _, units, _ in spans)
    # disarmed: nothing is recorded
    ops.encode(x, table.detach(), pm, None, 16, True, True)
    assert ops.timer.records is None


@pytest.mark.parametrize("n_in,n_out,nn,nl,grouped", [(80, 33, 64, 2, False), (112, 8, 64, 2, False), (80, 33, 32, 1, False),
                                                       (80, 1, 64, 2, False), (80, 33, 64, 2, True), (32, 32, 32, 2, False)])
def test_mlp_prepared_images_are_bit_identical(n_in, n_out, nn, nl, grouped):
    """DNS_MLP_PREPARED: dns_mlp_prepare writes the operand images the kernels' prologue would build (once per weight set); the
    forward and the backward (dX, dW, with and without dX) then COPY them in -- every output bit-identical."""
    import ctypes as C
    ops = _ops()
    from dns_slam_amd._lib import check, ptr, stream_ptr
    lib = ops.lib
    g = torch.Generator().manual_seed(17)
    P, G = 3000, (3 if grouped else 1)
    count = ops.mlp_param_count(n_in, n_out, nn, nl)
    params = (torch.randn(G, count, generator=g) * 0.2).to(DEV)
    x = torch.randn(P, n_in, generator=g).to(DEV)
    dy = torch.randn(P, n_out, generator=g).to(DEV)
    ri = tg = None
    n_slots = P
    if grouped:
        slot = torch.randint(0, G, (P,), generator=g).to(DEV)
        ri, tg, n_slots = ops.group_slots(slot, G, 2)
    nf = int(lib.dns_mlp_prepared_floats(n_in, n_out, nn, nl))
    assert nf > 0 and nf % 4 == 0
    prep = torch.empty(G, nf, device=DEV)
    check(lib.dns_mlp_prepare(ptr(params), n_in, n_out, nn, nl, G, count, ptr(prep), 0, stream_ptr()), "dns_mlp_prepare")
    stride = count if grouped else 0
    ws = torch.empty(int(lib.dns_mlp_bwd_ws_floats(n_slots, nn, nl)), device=DEV)
    outs = []
    for prepared in (False, True):
        w, flag = (prep, ops.MLP_PREPARED_FLAG) if prepared else (params, 0)
        y = torch.zeros(P, n_out, device=DEV)
        check(lib.dns_mlp_fwd(ptr(x), n_in, None, 0, 0, ptr(w), n_in, n_out, nn, nl, ptr(y), n_out, n_slots, ptr(ri), ptr(tg), stride,
                              None, flag, stream_ptr()), "dns_mlp_fwd")
        res = [y]
        for need_dx in (True, False):
            dx = torch.zeros(P, n_in, device=DEV) if need_dx else None
            dp = torch.zeros_like(params)
            check(lib.dns_mlp_bwd(ptr(x), n_in, None, 0, 0, ptr(dy), n_out, ptr(w), n_in, n_out, nn, nl, ptr(dx), n_in, None, 0, ptr(dp),
                                  ptr(ws), n_slots, ptr(ri), ptr(tg), stride, None, flag, stream_ptr()), "dns_mlp_bwd")
            res += [dx, dp] if need_dx else [dp]
        outs.append(res)
    torch.cuda.synchronize()
    assert float(outs[0][0].abs().max()) > 0 and float(outs[0][1].abs().max()) > 0
    for a, b, name in zip(outs[0], outs[1], ("y", "dx", "dW (with dx)", "dW (without dx)")):
        if name.startswith("dW"):                 # float atomics across workgroups: order-dependent sums, not bit-stable run to run
            assert_close(b.cpu(), a.cpu(), rtol=1e-5, elementwise=False, what=f"prepared images: {name}")
        else:
            assert torch.equal(a, b), name


def test_mlp_dwin_alone_equals_the_backward_with_it():
    """DNS_MLP_NO_DWIN + dns_mlp_dwin (the streaming dW_in = dH_1^T x kernel launched by the caller, e.g. on another stream)
    against dns_mlp_bwd launching both: same dX, same weight gradients (two-segment input, accumulate flags)."""
    ops = _ops()
    from dns_slam_amd._lib import check, ptr, stream_ptr
    lib = ops.lib
    g = torch.Generator().manual_seed(23)
    P, n1, n2, n_out, nn, nl = 5000, 48, 64, 8, 64, 2
    n_in = n1 + n2
    count = ops.mlp_param_count(n_in, n_out, nn, nl)
    params = (torch.randn(count, generator=g) * 0.2).to(DEV)
    x1, x2 = torch.randn(P, 80, generator=g).to(DEV), torch.randn(P, n2, generator=g).to(DEV)
    dy = torch.randn(P, n_out, generator=g).to(DEV)
    ws = torch.empty(int(lib.dns_mlp_bwd_ws_floats(P, nn, nl)), device=DEV)
    res = []
    for split in (False, True):
        d1, d2, dp = torch.zeros(P, 80, device=DEV), torch.zeros(P, n2, device=DEV), torch.zeros(count, device=DEV)
        flag = ops.MLP_NO_DWIN_FLAG if split else 0
        check(lib.dns_mlp_bwd(ptr(x1), 80, ptr(x2), n2, n1, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(d1), 80, ptr(d2), n2,
                              ptr(dp), ptr(ws), P, None, None, 0, None, 3 | flag, stream_ptr()), "dns_mlp_bwd")
        if split:
            w_in = dp[:nn * n_in].clone()
            assert float(w_in.abs().max()) == 0.0                      # the first layer's block is untouched until ...
            check(lib.dns_mlp_dwin(ptr(x1), 80, ptr(x2), n2, n1, n_in, nn, nl, ptr(dp), ptr(ws), P, None, None, 0, 0, stream_ptr()),
                  "dns_mlp_dwin")
        res.append((d1, d2, dp))
    torch.cuda.synchronize()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert float(res[1][2][:nn * n_in].abs().max()) > 0
    assert_close(res[1][2].cpu(), res[0][2].cpu(), rtol=1e-5, elementwise=False, what="dW with dns_mlp_dwin launched separately")


@pytest.mark.parametrize("nn,nl,n_out,two", [(64, 2, 3, True), (32, 1, 8, True), (64, 2, 33, False)])
def test_mlp_live_input_columns_equal_zero_padded_rows(nn, nl, n_out, two):
    """DNS_MLP_LIVE_IN(n): the trailing input columns [n, n_in) are identically zero (the reference's colour / logit networks when
    no 2-D feature code is attached, slams/mapping.py:553-557).  The kernels run as an n-input network on the same parameter
    tensor: against the full-width call on zero-padded rows -- forward, dX of the live columns, the dH_1 workspace, all weight
    gradients (dW_in: zero in the dead columns) -- to fp32 rounding."""
    import ctypes as C
    ops = _ops()
    from dns_slam_amd._lib import check, ptr, stream_ptr
    lib = ops.lib
    g = torch.Generator().manual_seed(31)
    P = 3000
    n_in, n1 = (112, 48) if two else (96, 0)
    live = 80
    count = ops.mlp_param_count(n_in, n_out, nn, nl)
    params = (torch.randn(count, generator=g) * 0.2).to(DEV)
    x_full = torch.randn(P, n_in, generator=g)
    x_full[:, live:] = 0.0
    dy = torch.randn(P, n_out, generator=g).to(DEV)
    if two:
        a_full = torch.zeros(P, 80)
        a_full[:, :n1] = x_full[:, :n1]
        b_full, b_live = x_full[:, n1:].contiguous(), x_full[:, n1:live].contiguous()
        xa, xb_full, xb_live = a_full.to(DEV), b_full.to(DEV), b_live.to(DEV)
    else:
        xa, xa_live = x_full.to(DEV), x_full[:, :live].contiguous().to(DEV)
    outs = []
    for use_live in (False, True):
        flag = ops.MLP_LIVE_IN(live) if use_live else 0
        y = torch.zeros(P, n_out, device=DEV)
        ws = torch.zeros(P * nn, device=DEV)
        dp = torch.zeros_like(params)
        if two:
            xb = xb_live if use_live else xb_full
            dxa, dxb = torch.zeros(P, 80, device=DEV), torch.zeros_like(xb)
            check(lib.dns_mlp_fwd(ptr(xa), 80, ptr(xb), xb.shape[1], n1, ptr(params), n_in, n_out, nn, nl, ptr(y), n_out, P, None, None, 0,
                                  None, flag, stream_ptr()), "fwd")
            check(lib.dns_mlp_bwd(ptr(xa), 80, ptr(xb), xb.shape[1], n1, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(dxa), 80,
                                  ptr(dxb), xb.shape[1], ptr(dp), ptr(ws), P, None, None, 0, None, flag, stream_ptr()), "bwd")
            res = [y, dxa[:, :n1], dxb[:, :live - n1], ws, dp]
        else:
            x = xa_live if use_live else xa
            dx = torch.zeros_like(x)
            check(lib.dns_mlp_fwd(ptr(x), x.shape[1], None, 0, 0, ptr(params), n_in, n_out, nn, nl, ptr(y), n_out, P, None, None, 0, None,
                                  flag, stream_ptr()), "fwd")
            check(lib.dns_mlp_bwd(ptr(x), x.shape[1], None, 0, 0, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(dx), x.shape[1],
                                  None, 0, ptr(dp), ptr(ws), P, None, None, 0, None, flag, stream_ptr()), "bwd")
            res = [y, dx[:, :live], ws, dp]
        outs.append(res)
    torch.cuda.synchronize()
    for a, b in zip(outs[1], outs[0]):
        assert float(b.abs().max()) > 0
        assert_close(a.cpu(), b.cpu(), rtol=1e-5, elementwise=False, what="live input columns vs zero-padded rows")
    w_in = outs[1][-1][:nn * n_in].reshape(nn, n_in)
    assert float(w_in[:, live:].abs().max()) == 0.0 and float(w_in[:, :live].abs().max()) > 0


def test_mlp_bwd_dx_from_skips_the_leading_columns():
    """DNS_MLP_DX_FROM(c): input-gradient columns [0, c) are neither formed nor stored (the smoothness lattice's coarse network needs
    the grid columns only); the others, the workspace and the weight gradients are what the plain call gives."""
    ops = _ops()
    from dns_slam_amd._lib import check, ptr, stream_ptr
    lib = ops.lib
    g = torch.Generator().manual_seed(41)
    P, n_in, n_out, nn, nl = 2500, 80, 1, 64, 2
    params = (torch.randn(ops.mlp_param_count(n_in, n_out, nn, nl), generator=g) * 0.2).to(DEV)
    x = torch.randn(P, n_in, generator=g).to(DEV)
    dy = torch.randn(P, n_out, generator=g).to(DEV)
    res = []
    for flag in (0, ops.MLP_DX_FROM(48)):
        dx = torch.full((P, n_in), -7.0, device=DEV)
        ws, dp = torch.zeros(P * nn, device=DEV), torch.zeros_like(params)
        check(lib.dns_mlp_bwd(ptr(x), n_in, None, 0, 0, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(dx), n_in, None, 0, ptr(dp),
                              ptr(ws), P, None, None, 0, None, flag, stream_ptr()), "bwd")
        res.append((dx, ws, dp))
    torch.cuda.synchronize()
    (dx0, ws0, dp0), (dx1, ws1, dp1) = res
    assert torch.equal(dx1[:, 48:], dx0[:, 48:]) and bool((dx1[:, :48] == -7.0).all()) and torch.equal(ws0, ws1)
    assert_close(dp1.cpu(), dp0.cpu(), rtol=1e-5, elementwise=False, what="DX_FROM: dW")


def test_render_nets_forward_only_reads_the_latent_out_of_the_fine_rows():
    """Forward-only callers without a 2-D code (the frame render) pass a code of ZERO columns: the colour / logit networks run as
    their live (OneBlob + latent)-input networks and read the latent as a column slice of the fine decoders' [P, 33] rows
    (dns_mlp_fwd: x2 of any 4-byte alignment, ldx2 = 33) -- no packed [P, 32] copy.  Same numbers as the full-width networks on
    an explicit code of zeros (up to the live-width operand scale of W_in: fp32 rounding)."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    P, G, pe_dim, hid, C, n_class, nn, nl = 5000, 3, 48, 32, 32, 8, 64, 2
    shp_c, shp_f = (80, hid + 1, nn, nl), (80, hid + 1, nn, nl)
    shp_col, shp_log = (pe_dim + hid + C, 3, nn, nl), (pe_dim + hid + C, n_class, nn, nl)
    cp = tr.mlp_init(*shp_c, g).to(DEV)
    pool = torch.stack([tr.mlp_init(*shp_f, g) for _ in range(G)]).to(DEV)
    colp, logp = tr.mlp_init(*shp_col, g).to(DEV), tr.mlp_init(*shp_log, g).to(DEV)
    buf = torch.randn(P, 80, generator=g).to(DEV)
    slot = torch.randint(0, G, (P,), generator=g).to(DEV)
    with torch.no_grad():
        _, f0, raw0, log0 = ops.render_nets(buf, torch.empty(P, 0, device=DEV), cp, pool, colp, logp, slot, pe_dim, shp_c, shp_f,
                                            shp_col, shp_log, need_coarse=False)
        _, f1, raw1, log1 = ops.render_nets(buf, torch.zeros(P, C, device=DEV), cp, pool, colp, logp, slot, pe_dim, shp_c, shp_f,
                                            shp_col, shp_log)
    assert torch.equal(f0, f1)
    assert_close(raw0.cpu(), raw1.cpu(), rtol=1e-6, what="raw, latent slice vs packed block of zeros")
    assert_close(log0.cpu(), log1.cpu(), rtol=1e-6, what="logits, latent slice vs packed block of zeros")


# ----------------------------------------------------------------------------------------- round 5
def test_oneblob_alone_with_64_bins_leaves_through_the_lds_tile():
    """A OneBlob-only call takes the LDS-tiled store path; with n_bins >= 43 its tile is larger than the 64 KB a kernel gets without
    the MaxDynamicSharedMemorySize attribute (such a call failed at launch)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    x = torch.rand(3000, 3, generator=g)
    xp = x.to(DEV).requires_grad_(True)
    y = ops.encode(xp, None, None, None, 64, True, False)
    assert_close(y.detach().cpu(), tr.oneblob_forward(x, 64), rtol=1e-6, what="oneblob n=64")
    # ... and its backward (OneBlob-only gradients come in through the LDS tile too when it fits 64 KB: 16 bins; 64 bins: direct rows)
    gy = torch.randn(3000, 192, generator=g)
    (y * gy.to(DEV)).sum().backward()
    xo = x.clone().requires_grad_(True)
    (tr.oneblob_forward(xo, 64) * gy).sum().backward()
    d = (xp.grad.cpu() - xo.grad).abs()
    assert float((d > 1e-4 * xo.grad.abs().max()).float().mean()) < 0.002


def test_pair_list_scatter_of_tiny_gradients_stays_finite(monkeypatch):
    """The pair-list bins are fixed-point with a scale 2^(40 - ex) from the largest gradient: below 2^-87 that scale left fp32's
    range (inf / NaN in the table gradient).  Gradients of ~1e-30 must come out as (tiny) finite numbers equal to
    1e-30 times the gradient of the unscaled problem."""
    ops = _ops()
    pm = ops.GridMeta(16, 592)
    g = torch.Generator().manual_seed(13)
    x = torch.rand(20000, 3, generator=g).to(DEV)
    gy = torch.randn(20000, 32, generator=g).to(DEV)
    monkeypatch.setattr(ops, "SCATTER_FORM", (ops.SCATTER_LISTS, 0))
    out = []
    for s in (1.0, 1e-30):
        table = torch.rand(pm.total_rows * 2, generator=torch.Generator().manual_seed(1)).to(DEV).requires_grad_(True)
        ops.encode(x, table, pm, None, 16, False, True).backward(gy * s)
        out.append(table.grad.clone())
    assert bool(torch.isfinite(out[1]).all())
    ref = out[0].double() * 1e-30
    assert float((out[1].double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
