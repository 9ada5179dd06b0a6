"""Half rows (ABI v12, BASELINE configs[4]): the MLP kernels in tcnn's OWN arithmetic -- f16 activations and weight operands, fp32
accumulation, a static loss scale (reference models/decoder.py:58-64,84-90,94,101-116) -- against a torch
restatement of exactly that arithmetic (every operand rounded to f16 where the kernel rounds it, products and sums in float64),
plus the f16 row writers (encoder, feature block) against the fp32 ones."""
import ctypes as C

import pytest
import torch

from oracle import tcnn_ref as tr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
S128 = 128.0


def _ops():
    from dns_slam_amd import ops
    return ops


def _split(w, n_in, n_out, nn, nl, n_in_w=None):
    n_in_w = n_in_w or n_in
    o, Ws = 0, []
    for r, c in [(nn, n_in_w)] + [(nn, nn)] * (nl - 1) + [(n_out, nn)]:
        Ws.append(w[o:o + r * c].reshape(r, c))
        o += r * c
    Ws[0] = Ws[0][:, :n_in]                       # the live columns
    return Ws, o


def _emulate(x16, w, dy, n_in, n_out, nn, nl, scale=S128, n_in_w=None):
    """What the half-rows kernels compute, restated with torch.  x16: [P, n_in] float16.  Returns y, dx, [dW per matrix]."""
    q = lambda t: t.to(torch.float16).to(torch.float64)          # round to f16 (subnormals kept), exact afterwards
    Ws, _ = _split(w, n_in, n_out, nn, nl, n_in_w)
    acts = [x16.double()]
    for W in Ws[:-1]:
        acts.append(q(torch.relu((acts[-1] @ q(W).T).float())))   # fp32 accumulator -> ReLU -> f16
    y = (acts[-1] @ q(Ws[-1]).T).float()
    g = q(dy.float() * scale)                                       # S dY, f16
    dWs = [None] * len(Ws)
    for li in range(len(Ws) - 1, -1, -1):
        dWs[li] = ((g.T @ acts[li]) / scale).float()
        d_in = (g @ q(Ws[li])).float()                              # fp32 accumulator
        if li > 0:
            g = q(d_in) * (acts[li] > 0).double()                   # ReLU' on the f16 activation, then f16
        else:
            dx = d_in / scale
    return y, dx, dWs


def _rms_rel(a, b):
    return float((a.double() - b.double()).pow(2).mean().sqrt()) / max(float(b.double().pow(2).mean().sqrt()), 1e-30)


def _check(name, got, want, tol=2e-4):
    """Kernel vs emulation: the two differ by the ORDER of fp32 sums only (typical relative rms 1e-5); 2e-4 leaves room for the
    occasional 1-ulp flip of an f16 rounding behind such a sum."""
    r = _rms_rel(got, want)
    assert r <= tol, f"{name}: relative rms error {r:.3e} > {tol}"


def _relu_flip_rows(dx, dxe, max_frac=1e-3):
    """Rows whose input gradient is far off: a hidden unit whose fp32 pre-activation is ~0 may fall on the other side of the ReLU
    (the sums' order differs), which changes that POINT's gradients wholesale -- and, through it, one rank-1 term of every
    weight gradient.  The tests zero dY of these rows (every row's arithmetic is independent of the others) and compare again."""
    err = (dx.double() - dxe.double()).abs().max(1)[0]
    rows = (err > 1e-3 * float(dxe.abs().max())).nonzero().reshape(-1)
    assert rows.numel() <= max(1, int(max_frac * dx.shape[0])), f"{rows.numel()} of {dx.shape[0]} rows far off: more than ReLU flips explain"
    return rows


@pytest.mark.parametrize("n_in,n_out,nn,nl,P", [(80, 33, 64, 2, 5000), (80, 33, 32, 1, 4097), (112, 8, 64, 2, 3000), (32, 3, 64, 1, 1000),
                                                  (128, 64, 32, 2, 2049), (48, 16, 64, 2, 777), (80, 1, 64, 2, 40000)])
def test_mlp_half_forward_backward_vs_f16_emulation(n_in, n_out, nn, nl, P):
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    w = tr.mlp_init(n_in, n_out, nn, nl, g).to(DEV) * 3.0
    x16 = torch.randn(P, n_in, generator=g).to(DEV).half()
    dy = (torch.randn(P, n_out, generator=g) * 1e-3).to(DEV)
    y = ops.mlp_fwd_half(x16, w, n_in, n_out, nn, nl)
    _, used = _split(w, n_in, n_out, nn, nl)

    def run():
        dx = torch.full((P, n_in), float("nan"), device=DEV)
        dw = torch.zeros_like(w)
        ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_x=dx, d_params=dw)
        return dx, dw, _emulate(x16, w, dy, n_in, n_out, nn, nl)

    dx, dw, (ye, dxe, dWe) = run()
    _check("y", y, ye)
    flips = _relu_flip_rows(dx, dxe)
    if flips.numel():
        dy[flips] = 0.0
        dx, dw, (ye, dxe, dWe) = run()
    _check("dx", dx, dxe)
    _check("d_params", dw[:used], torch.cat([t.reshape(-1) for t in dWe]))
    assert float(dw[used:].abs().max() if used < dw.numel() else 0.0) == 0.0, "padding rows of W_out received a gradient"
    # the frozen-scene form (no weight gradients) gives the same input gradient
    dx2 = torch.empty_like(dx)
    ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_x=dx2)
    assert torch.equal(dx, dx2)
    # weight gradients without an input gradient
    dw2 = torch.zeros_like(w)
    ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_params=dw2)
    _check("d_params (no dx)", dw2[:used], dw[:used], tol=1e-5)   # float atomics across workgroups: not bit-reproducible


def test_mlp_half_loss_scale_is_exact_power_of_two_bookkeeping():
    """The loss scale only moves the f16 window: with gradients well inside f16's range at both scales the results agree to
    fp32 rounding; gradients of 1e-7 (below f16's normal range at scale 1) survive at scale 128 -- what the scale is for."""
    ops = _ops()
    n_in, n_out, nn, nl, P = 80, 33, 64, 2, 3000
    g = torch.Generator().manual_seed(3)
    w = tr.mlp_init(n_in, n_out, nn, nl, g).to(DEV) * 3.0
    x16 = torch.randn(P, n_in, generator=g).to(DEV).half()
    dy = (torch.randn(P, n_out, generator=g) * 1e-2).to(DEV)

    def run(dy_, s):
        dx, dw = torch.empty(P, n_in, device=DEV), torch.zeros_like(w)
        ops.mlp_bwd_half(x16, dy_, w, n_in, n_out, nn, nl, d_x=dx, d_params=dw, loss_scale=s)
        return dx, dw

    a, b = run(dy, 128.0), run(dy, 16.0)
    assert _rms_rel(a[0], b[0]) <= 1e-3 and _rms_rel(a[1], b[1]) <= 1e-3
    tiny = run(dy * 1e-5, 128.0)                         # ~1e-7: subnormal in f16 without the scale
    assert _rms_rel(tiny[0] / 1e-5, a[0]) <= 3e-2 and _rms_rel(tiny[1] / 1e-5, a[1]) <= 3e-2
    with pytest.raises(ValueError):
        run(dy, 0.0)


@pytest.mark.parametrize("nn,nl,n_out,n2,live", [(64, 2, 3, 32, 0), (64, 2, 8, 64, 0), (32, 1, 8, 32, 0), (64, 2, 3, 32, 80), (64, 1, 8, 32, 80)])
def test_mlp_half_two_segment_input(nn, nl, n_out, n2, live):
    """The colour / logit networks' input (models/decoder.py:123-125: torch.cat((pe, features), -1)) as two f16 row sets, with
    the second segment's gradient ADDED into a strided view and -- DNS_MLP_LIVE_IN -- a network whose parameter tensor is wider
    than its live input (no 2-D code: slams/mapping.py:553-557)."""
    ops = _ops()
    n1, P = 48, 3000
    n_live = n1 + n2
    n_in_w = 112 if live else n_live                 # row stride of W_in in the parameter tensor
    g = torch.Generator().manual_seed(11)
    w = tr.mlp_init(n_in_w, n_out, nn, nl, g).to(DEV) * 3.0
    xa = torch.randn(P, 80, generator=g).to(DEV).half()          # [pe | grid] rows: the network reads the first 48 columns
    xb = torch.randn(P, n2, generator=g).to(DEV).half()
    dy = (torch.randn(P, n_out, generator=g) * 1e-3).to(DEV)
    x1 = xa[:, :n1]
    y = ops.mlp_fwd_half(x1, w, n_in_w, n_out, nn, nl, x2=xb, live_in=n_live if live else 0)
    xcat = torch.cat((x1, xb), -1)
    ye, dxe, dWe = _emulate(xcat, w, dy, n_live, n_out, nn, nl, n_in_w=n_in_w)
    _check("y", y, ye)
    acc = 2 | (ops.MLP_LIVE_IN(n_live) if live else 0)            # d_x overwritten, d_x2 +=

    def run():
        d1 = torch.full((P, 80), 7.0, device=DEV)
        d2x = torch.full((P, 4 + n2), 5.0, device=DEV)
        dw = torch.zeros_like(w)
        ops.mlp_bwd_half(x1, dy, w, n_in_w, n_out, nn, nl, x2=xb, d_x=d1, d_x2=d2x[:, 4:], d_params=dw, accumulate=acc)
        return d1, d2x, dw, _emulate(xcat, w, dy, n_live, n_out, nn, nl, n_in_w=n_in_w)

    d1, d2x, dw, (ye, dxe, dWe) = run()
    flips = _relu_flip_rows(torch.cat((d1[:, :n1], d2x[:, 4:] - 5.0), -1), dxe)
    if flips.numel():
        dy[flips] = 0.0
        d1, d2x, dw, (ye, dxe, dWe) = run()
    _check("d_x", d1[:, :n1], dxe[:, :n1])
    assert float((d1[:, n1:] - 7.0).abs().max()) == 0.0, "d_x written beyond its segment"
    _check("d_x2 (+=)", d2x[:, 4:] - 5.0, dxe[:, n1:], tol=2e-3)   # (the sum with 5.0 costs fp32 bits of a ~1e-3 gradient)
    assert float((d2x[:, :4] - 5.0).abs().max()) == 0.0
    Win = torch.zeros(nn, n_in_w, device=DEV)
    Win[:, :n_live] = dWe[0]
    want = torch.cat([Win.reshape(-1)] + [t.reshape(-1) for t in dWe[1:]])
    _check("d_params", dw[:want.numel()], want)
    # DNS_MLP_DX_FIRST: only the first segment's input gradient
    d1b = torch.empty(P, n1, device=DEV)
    ops.mlp_bwd_half(x1, dy, w, n_in_w, n_out, nn, nl, x2=xb, d_x=d1b, accumulate=ops.MLP_DX_FIRST_FLAG | (ops.MLP_LIVE_IN(n_live) if live else 0))
    assert torch.equal(d1b, d1[:, :n1])


def test_mlp_half_grouped_matches_per_class_loop_and_dx_from():
    """Per-class fine decoders (slams/mapping.py:590-601) through row_index / tile_group, read-add-write of the shared input
    gradient, and DNS_MLP_DX_FROM (the lattice's network: grid columns only)."""
    ops = _ops()
    n_in, n_out, nn, nl, P, G = 80, 33, 64, 2, 6000, 5
    g = torch.Generator().manual_seed(5)
    stride = (tr.mlp_init(n_in, n_out, nn, nl, g).numel() + 3) // 4 * 4
    pool = torch.zeros(G, stride)
    for k in range(G):
        wk = tr.mlp_init(n_in, n_out, nn, nl, g) * 3.0
        pool[k, :wk.numel()] = wk
    pool = pool.to(DEV)
    x16 = torch.randn(P, n_in, generator=g).to(DEV).half()
    dy = (torch.randn(P, n_out, generator=g) * 1e-3).to(DEV)
    slot = torch.randint(-1, G, (P,), generator=g).to(DEV)
    slot[slot == 3] = -1                                   # an empty weight set
    ri, tg, n_slots = ops.group_slots(slot, G, 2)
    y = torch.zeros(P, n_out, device=DEV)
    ops.mlp_fwd_half(x16, pool, n_in, n_out, nn, nl, row_index=ri, tile_group=tg, n_slots=n_slots, param_stride=stride, out=y)
    dx = torch.full((P, n_in), 1.0, device=DEV)
    dpool = torch.zeros_like(pool)
    ops.mlp_bwd_half(x16, dy, pool, n_in, n_out, nn, nl, d_x=dx, d_params=dpool, row_index=ri, tile_group=tg, n_slots=n_slots,
                     param_stride=stride, accumulate=1)
    for k in range(G):
        idx = (slot == k).nonzero().reshape(-1)
        if idx.numel() == 0:
            assert float(dpool[k].abs().max()) == 0.0
            continue
        ye, dxe, dWe = _emulate(x16[idx], pool[k], dy[idx], n_in, n_out, nn, nl)
        _check(f"y class {k}", y[idx], ye)
        flips = _relu_flip_rows(dx[idx] - 1.0, dxe, max_frac=2e-3)
        keep = torch.ones(idx.numel(), dtype=torch.bool, device=DEV)
        keep[flips] = False
        _check(f"dx class {k}", (dx[idx] - 1.0)[keep], dxe[keep], tol=2e-3)   # (the sum with 1.0 costs fp32 bits of a ~1e-3 gradient)
        want = torch.cat([t.reshape(-1) for t in dWe])
        _check(f"d_params class {k}", dpool[k, :want.numel()], want, tol=2e-4 if flips.numel() == 0 else 2e-2)
    none = (slot < 0).nonzero().reshape(-1)
    assert float(y[none].abs().max()) == 0.0 and float((dx[none] - 1.0).abs().max()) == 0.0
    # DX_FROM(48): the OneBlob columns' gradient is neither formed nor stored
    w = pool[0, :].contiguous()
    dxa, dxb = torch.full((P, n_in), 9.0, device=DEV), torch.empty(P, n_in, device=DEV)
    ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_x=dxa, accumulate=ops.MLP_DX_FROM(48))
    ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_x=dxb)
    assert float((dxa[:, :32] - 9.0).abs().max()) == 0.0          # whole tiles below the first live column are skipped
    assert torch.equal(dxa[:, 48:], dxb[:, 48:])


def test_half_row_writers_round_the_fp32_rows():
    """dns_encode_fwd_split / dns_feature_block_split with DNS_SPLIT_PLAIN write exactly f16(fp32 row)."""
    ops = _ops()
    from dns_slam_amd._lib import check, ptr, stream_ptr
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    bound, cam, frames = synthetic.make_scene(2, seed=1)
    cfg = synthetic.default_cfg(n_pixels=64, hash_size=14, voxel_size=0.08)
    dec = Decoder(cfg["model"], bound, n_class=8).to(DEV)
    with torch.no_grad():
        dec.pe_fn.grid_fn.params.uniform_(-1.0, 1.0)
    gm = dec.pe_fn.grid_fn
    P, n_bins = 3001, dec.pe_fn.pe_fn.n_bins
    pe, gd = 3 * n_bins, dec.grid_dim
    ld = pe + gd
    g = torch.Generator().manual_seed(2)
    pts = torch.rand(P, 3, generator=g).to(DEV)
    f32 = torch.empty(P, ld, device=DEV)
    check(ops.lib.dns_encode_fwd(ptr(pts), None, P, n_bins, ptr(gm.params), C.byref(gm.meta.c), None, ptr(f32), ld,
                                 C.c_void_p(f32.data_ptr() + 4 * pe), ld, None, stream_ptr()), "dns_encode_fwd")
    xh = torch.empty(P, ld, device=DEV, dtype=torch.float16)
    check(ops.lib.dns_encode_fwd_split(ptr(pts), None, P, n_bins, ptr(gm.params), C.byref(gm.meta.c), None, None, 0, ptr(xh), ld,
                                       None, ops.SPLIT_PLAIN, None, stream_ptr()), "dns_encode_fwd_split")
    assert torch.equal(xh, f32.half())
    # feature block
    N, S, hid, Cc = 50, 60, 32, 32
    fine = torch.randn(N * S, hid + 1, generator=g).to(DEV)
    code = torch.randn(N * S, Cc, generator=g).to(DEV)
    z = (torch.rand(N, S, generator=g) * 4).to(DEV)
    gd_ = (torch.rand(N, generator=g) * 4).to(DEV)
    for cd, Cn in ((code, Cc), (None, 0)):
        F = hid + Cn
        feat, raw = torch.empty(N * S, F, device=DEV), torch.zeros(N * S, 4, device=DEV)
        check(ops.lib.dns_feature_block(ptr(fine), hid + 1, hid, ptr(cd), Cn, ptr(z), ptr(gd_), N, S, ptr(feat), F, ptr(raw),
                                        stream_ptr()), "dns_feature_block")
        fh, raw2 = torch.empty(N * S, F, device=DEV, dtype=torch.float16), torch.zeros(N * S, 4, device=DEV)
        check(ops.lib.dns_feature_block_split(ptr(fine), hid + 1, hid, ptr(cd), Cn, 1, 0, ptr(z), ptr(gd_), N, S, None, 0, ptr(fh), F,
                                              None, ops.SPLIT_PLAIN, ptr(raw2), stream_ptr()), "dns_feature_block_split")
        assert torch.equal(fh, feat.half()) and torch.equal(raw, raw2)


def test_map_step_on_half_rows_tracks_the_fp16_operand_and_fp32_steps(monkeypatch):
    """fused_step.MapStep with fp16 networks (BASELINE configs[4]) on the half-rows kernels: on the SAME draws its loss terms and
    every gradient segment agree with the same step on round 4's fp16-operand kernels (both round operands to f16; the scale
    bookkeeping differs: static 128 vs per-point) and with the fp32-grade step to the stated, looser tolerance; it trains."""
    from dns_slam_amd import synthetic
    from dns_slam_amd.decoder import Decoder
    from dns_slam_amd.mapping import Mapper
    from dns_slam_amd.fused_step import MapStep
    from util import randomise_
    res = {}
    for mode in ("fp32", "operand", "half"):
        monkeypatch.setenv("DNS_HALF_ROWS", "1" if mode == "half" else "0")
        cam = synthetic.camera(H=60, W=80, fx=60.0, fy=60.0)
        bound, cam, frames = synthetic.make_scene(4, cam=cam, seed=0)
        cfg = synthetic.default_cfg(n_pixels=360, n_samples_ray=32, n_surface_ray=15, n_frames=4, hash_size=14, voxel_size=0.08,
                                    n_neurons=64, n_hidden_layers=2, smooth_pts=12, mlp_dtype="fp32" if mode == "fp32" else "fp16")
        dec = Decoder(cfg["model"], bound, n_class=8).to(DEV)
        mapper = Mapper(cfg, dec, bound, cam, device=DEV)
        mapper.set_decoder(frames)
        randomise_(dec, 11, scale=1.0)
        with torch.no_grad():
            dec.pe_fn.grid_fn.params.mul_(2000.0)
        randomise_([mapper.fine_decoders.pool], 12)
        mapper.static_shapes, mapper.is_BA, mapper.overlap_smooth, mapper.prefetch_draws = True, True, True, True
        _, ql, Tl = mapper.set_optimizer(frames, fused=True)
        ms = MapStep(mapper, frames, ql, Tl)
        assert ms.half == (mode == "half") and bool(ms.fp16) == (mode != "fp32")
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        ms.step()
        torch.cuda.synchronize()
        names = ("color", "logit", "pool", "table", "coarse", "quat", "trans")
        first = {"loss": float(ms.losses()[0]), "terms": {k: float(v) for k, v in ms.losses()[1].items()},
                 "g": {n: getattr(ms.cur, "g_" + n).clone() for n in names}}
        for _ in range(11):
            ms.step()
        res[mode] = (first, float(ms.losses()[0]))
    rel = lambda a, b: float((a.double() - b.double()).pow(2).mean().sqrt() / b.double().pow(2).mean().sqrt().clamp_min(1e-30))
    h, o, f = res["half"][0], res["operand"][0], res["fp32"][0]
    assert abs(h["loss"] - o["loss"]) <= 2e-3 * abs(o["loss"]) and abs(h["loss"] - f["loss"]) <= 2e-2 * abs(f["loss"])
    for k in h["terms"]:
        assert abs(h["terms"][k] - o["terms"][k]) <= 5e-3 * max(abs(o["terms"][k]), 1e-3), k
    for n in h["g"]:
        assert rel(h["g"][n], o["g"][n]) <= 2e-2, (n, rel(h["g"][n], o["g"][n]))
        assert rel(h["g"][n], f["g"][n]) <= 1e-1, (n, rel(h["g"][n], f["g"][n]))
    assert res["half"][1] < 0.9 * h["loss"], "the half-rows step does not train"
    assert abs(res["half"][1] - res["fp32"][1]) <= 0.1 * abs(res["fp32"][1])
