"""The split-row activation format (include/dns_hip.h "split rows", csrc/split_rows.hpp) between the encoders and the MLP kernels:
producers (dns_encode_fwd_split, dns_feature_block_split) against the fp32 entry points and the format's definition, consumers
(dns_mlp_fwd_split / dns_mlp_bwd_split) against dns_mlp_fwd / dns_mlp_bwd on the SAME values -- which the oracle tests of
test_gpu_kernels.py hold to oracle/tcnn_ref.py.  A one-segment network must give IDENTICAL results on either input form (the
exponent and the two halfs are what the fp32 kernels derive per launch); a two-segment input aligns two exponents and agrees to
fp32 rounding."""
import ctypes as C

import pytest
import torch

from oracle import tcnn_ref as tr
from util import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _lib():
    from dns_slam_amd import ops
    from dns_slam_amd._lib import DnsSplitRows, check, ptr, stream_ptr
    return ops, ops.lib, DnsSplitRows, check, ptr, stream_ptr


def _scale_exp(m):
    """csrc/split_rows.hpp scale_exp: m 2^e in [2^13, 2^14); 0 for m = 0."""
    _, ex = torch.frexp(m)
    return torch.where(m > 0, (14 - ex).clamp(-110, 110), torch.zeros_like(ex))


def _decode(xs, exps, K, lo=True):
    """[P, ld] halfs + [P] exponents -> fp32 [P, K]"""
    v = xs[:, :K].float()
    if lo:
        v = v + xs[:, K:2 * K].float()
    return torch.ldexp(v, -exps[:, None])


def _split_rows(x):
    """the format's definition in torch: e = scale_exp(max |row|), hi = f16(x 2^e), lo = f16(x 2^e - hi)"""
    e = _scale_exp(x.abs().amax(dim=1)).to(torch.int32)
    xs = torch.ldexp(x, e[:, None])
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return torch.cat((hi, lo), 1).contiguous(), e.contiguous()


def _rows(DnsSplitRows, t, e, K, lo=True):
    return DnsSplitRows(t.data_ptr(), e.data_ptr(), t.shape[1], K if lo else 0)


@pytest.mark.parametrize("hash_size,res,P,hi_only,world", [(16, 592, 5000, False, True), (12, 64, 777, False, False), (16, 592, 3001, True, True)])
def test_encode_fwd_split_equals_the_fp32_rows(hash_size, res, P, hi_only, world):
    ops, lib, DnsSplitRows, check, ptr, stream_ptr = _lib()
    pm = ops.GridMeta(hash_size, res)
    g = torch.Generator().manual_seed(2)
    table = (torch.rand(pm.total_rows, 2, generator=g) * 2 - 1)
    table[: pm.total_rows // 2] *= 1e-3                    # levels of very different magnitude inside one row
    x = torch.rand(P, 3, generator=g) * 1.1 - 0.05
    bound = torch.tensor([[-1.0, 2.0], [-2.0, 1.5], [0.0, 3.0]], dtype=torch.float64)
    pts = (x.double() * (bound[:, 1] - bound[:, 0]) + bound[:, 0]).float() if world else x
    b6 = ops._bound6(bound) if world else None
    pts_d, table_d = pts.to(DEV), table.to(DEV)
    K = 80
    ref = torch.empty(P, K, device=DEV)
    x_ref = torch.empty(P, 3, device=DEV)
    dydx_ref = torch.empty(16 * 3 * P * 2, device=DEV)
    check(lib.dns_encode_fwd(ptr(pts_d), b6, P, 16, ptr(table_d), C.byref(pm.c), ptr(x_ref), ptr(ref), K, C.c_void_p(ref.data_ptr() + 4 * 48),
                             K, ptr(dydx_ref), stream_ptr()), "dns_encode_fwd")
    f32 = torch.full((P, K), 7.0, device=DEV)
    ld = K if hi_only else 2 * K
    xs = torch.zeros(P, ld, device=DEV, dtype=torch.float16)
    ex = torch.empty(P, device=DEV, dtype=torch.int32)
    x_out = torch.empty(P, 3, device=DEV)
    dydx = torch.empty_like(dydx_ref)
    check(lib.dns_encode_fwd_split(ptr(pts_d), b6, P, 16, ptr(table_d), C.byref(pm.c), ptr(x_out), ptr(f32), K, ptr(xs), ld, ptr(ex),
                                   1 if hi_only else 0, ptr(dydx), stream_ptr()), "dns_encode_fwd_split")
    torch.cuda.synchronize()
    assert torch.equal(f32, ref) and torch.equal(x_out, x_ref) and torch.equal(dydx, dydx_ref)
    want_xs, want_e = _split_rows(ref.cpu())
    assert torch.equal(ex.cpu(), want_e)
    if hi_only:
        assert torch.equal(xs.cpu(), want_xs[:, :K])
    else:
        assert torch.equal(xs.cpu(), want_xs)
        # and the two halfs carry the fp32 value to 2^-22 of the row's maximum
        err = (_decode(xs.cpu(), ex.cpu(), K) - ref.cpu()).abs().amax(dim=1)
        assert bool((err <= 2.0 ** -21 * ref.cpu().abs().amax(dim=1)).all())
    # without the fp32 copy: the same split rows
    xs2 = torch.zeros_like(xs)
    check(lib.dns_encode_fwd_split(ptr(pts_d), b6, P, 16, ptr(table_d), C.byref(pm.c), None, None, 0, ptr(xs2), ld, ptr(ex),
                                   1 if hi_only else 0, None, stream_ptr()), "dns_encode_fwd_split")
    assert torch.equal(xs2, xs)


@pytest.mark.parametrize("with_code,n_ref,hi_only", [(True, 1, False), (False, 1, False), (True, 3, False), (True, 1, True)])
def test_feature_block_split_equals_the_fp32_block(with_code, n_ref, hi_only):
    ops, lib, DnsSplitRows, check, ptr, stream_ptr = _lib()
    g = torch.Generator().manual_seed(4)
    Kf, Npf, S, H, Cc = 2, 25, 9, 32, 32
    N = Kf * Npf
    P, Pi = N * S, Npf * S
    fine = torch.randn(P, H + 1, generator=g) * torch.logspace(-3, 2, P)[:, None]        # rows of very different magnitude
    code = torch.rand(Kf, n_ref, Pi, Cc, generator=g) * 2 - 1
    d = torch.rand(N, generator=g) * 3
    d[::7] = 0.0
    z = d[:, None] * (0.8 + 0.4 * torch.rand(N, S, generator=g)) + 0.01
    F = H + Cc
    ld = F if hi_only else 2 * F
    feat = torch.empty(P, F, device=DEV)
    xs = torch.zeros(P, ld, device=DEV, dtype=torch.float16)
    ex = torch.empty(P, device=DEV, dtype=torch.int32)
    raw = torch.zeros(P, 4, device=DEV)
    fine_d, code_d, z_d, d_d = fine.to(DEV), code.to(DEV), z.to(DEV), d.to(DEV)
    check(lib.dns_feature_block_split(ptr(fine_d), H + 1, H, ptr(code_d) if with_code else None, Cc, n_ref, Pi, ptr(z_d), ptr(d_d), N, S,
                                      ptr(feat), F, ptr(xs), ld, ptr(ex), 1 if hi_only else 0, ptr(raw), stream_ptr()),
          "dns_feature_block_split")
    torch.cuda.synchronize()
    dd = d[:, None]
    trunc = ((1.0 - (z < dd * 0.95).float()) * (1.0 - (z > dd * 1.05).float()) * (dd > 0.0).float()).reshape(P, 1)
    if with_code:
        merged = code.mean(dim=1) if n_ref > 1 else code[:, 0]                         # models/decoder.py:76
        want_code = merged.reshape(P, Cc) * trunc
    else:
        want_code = torch.zeros(P, Cc)
    want = torch.cat((fine[:, 1:], want_code), -1)
    if n_ref > 1:
        assert_close(feat.cpu(), want, rtol=1e-6, what="feature block, mean over references")
    else:
        assert torch.equal(feat.cpu(), want)
    assert torch.equal(raw.cpu()[:, 3], fine[:, 0])
    want_xs, want_e = _split_rows(feat.cpu())
    assert torch.equal(ex.cpu(), want_e)
    assert torch.equal(xs.cpu(), want_xs[:, :F] if hi_only else want_xs)


@pytest.mark.parametrize("n_in,n_out,nn,nl,grouped,fp16", [(80, 33, 64, 2, False, False), (80, 33, 32, 1, True, False),
                                                           (112, 8, 64, 2, False, False), (48, 3, 32, 2, False, False),
                                                           (80, 33, 64, 2, True, True), (128, 16, 64, 1, False, True)])
def test_mlp_on_split_rows_is_identical_to_fp32_rows(n_in, n_out, nn, nl, grouped, fp16):
    """One input segment: forward outputs, dX and the dH_1 workspace are bit-identical on either input form; the weight
    gradients (float atomics across workgroups) to 1e-5."""
    ops, lib, DnsSplitRows, check, ptr, stream_ptr = _lib()
    g = torch.Generator().manual_seed(17)
    P, G = 3000, (3 if grouped else 1)
    count = ops.mlp_param_count(n_in, n_out, nn, nl)
    params = (torch.randn(G, count, generator=g) * 0.2).to(DEV)
    x = (torch.randn(P, n_in, generator=g) * torch.logspace(-4, 3, P)[:, None]).to(DEV)
    x[5] = 0.0                                               # an all-zero row: exponent 0
    dy = torch.randn(P, n_out, generator=g).to(DEV)
    ri = tg = None
    n_slots = P
    if grouped:
        slot = torch.randint(0, G, (P,), generator=g).to(DEV)
        ri, tg, n_slots = ops.group_slots(slot, G, 2)
    stride = count if grouped else 0
    xs, ex = _split_rows(x.cpu())
    if fp16:
        xs = xs[:, :n_in].contiguous()                       # half-width rows: no lo plane
    xs, ex = xs.to(DEV), ex.to(DEV)
    rows = _rows(DnsSplitRows, xs, ex, n_in, lo=not fp16)
    flag = ops.MLP_FP16_FLAG if fp16 else 0
    ws = [torch.zeros(int(lib.dns_mlp_bwd_ws_floats(n_slots, nn, nl)), device=DEV) for _ in range(2)]
    y = [torch.zeros(P, n_out, device=DEV) for _ in range(2)]
    dx = [torch.zeros(P, n_in, device=DEV) for _ in range(2)]
    dp = [torch.zeros_like(params) for _ in range(2)]
    check(lib.dns_mlp_fwd(ptr(x), n_in, None, 0, 0, ptr(params), n_in, n_out, nn, nl, ptr(y[0]), n_out, n_slots, ptr(ri), ptr(tg), stride,
                          None, flag, stream_ptr()), "dns_mlp_fwd")
    check(lib.dns_mlp_fwd_split(C.byref(rows), None, 0, ptr(params), n_in, n_out, nn, nl, ptr(y[1]), n_out, n_slots, ptr(ri), ptr(tg),
                                stride, flag, stream_ptr()), "dns_mlp_fwd_split")
    check(lib.dns_mlp_bwd(ptr(x), n_in, None, 0, 0, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(dx[0]), n_in, None, 0, ptr(dp[0]),
                          ptr(ws[0]), n_slots, ptr(ri), ptr(tg), stride, None, flag | ops.MLP_NO_DWIN_FLAG, stream_ptr()), "dns_mlp_bwd")
    check(lib.dns_mlp_bwd_split(C.byref(rows), None, 0, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(dx[1]), n_in, None, 0,
                                ptr(dp[1]), ptr(ws[1]), n_slots, ptr(ri), ptr(tg), stride, flag, stream_ptr()), "dns_mlp_bwd_split")
    torch.cuda.synchronize()
    assert float(y[0].abs().max()) > 0 and float(dx[0].abs().max()) > 0
    assert torch.equal(y[0], y[1]), "forward"
    assert torch.equal(dx[0], dx[1]), "dX"
    assert torch.equal(ws[0], ws[1]), "dH_1 workspace"
    assert_close(dp[1].cpu(), dp[0].cpu(), rtol=1e-5, elementwise=False, what="split rows: dW")
    # the frozen-scene form (no weight gradients): dX alone
    dx2 = torch.zeros(P, n_in, device=DEV)
    check(lib.dns_mlp_bwd_split(C.byref(rows), None, 0, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(dx2), n_in, None, 0,
                                None, None, n_slots, ptr(ri), ptr(tg), stride, flag, stream_ptr()), "dns_mlp_bwd_split")
    assert torch.equal(dx2, dx[0]), "dX without weight gradients"


@pytest.mark.parametrize("nn,nl,n_out,fp16", [(64, 2, 3, False), (32, 1, 8, False), (64, 2, 8, True)])
def test_mlp_two_segment_split_rows(nn, nl, n_out, fp16):
    """Colour / logit network form: columns [0, 48) from the encoder's 80-wide rows, [48, 112) from the feature block -- two row
    sets with their own exponents, aligned in the kernel.  Against the fp32 two-segment entry points (1e-6 of the output scale:
    the alignment multiplies already-rounded halfs by a power of two, where the fp32 path rounds after scaling) and against the
    float64 oracle network (1e-4)."""
    ops, lib, DnsSplitRows, check, ptr, stream_ptr = _lib()
    g = torch.Generator().manual_seed(23)
    P, n1, n2 = 2500, 48, 64
    n_in = n1 + n2
    count = ops.mlp_param_count(n_in, n_out, nn, nl)
    params = (torch.randn(count, generator=g) * 0.2).to(DEV)
    enc = torch.randn(P, 80, generator=g)
    enc[:, :48] = enc[:, :48].abs().clamp(max=1.0)            # OneBlob-like
    feat = torch.randn(P, n2, generator=g) * torch.logspace(-3, 3, P)[:, None]      # from far below to far above the first segment
    dy = torch.randn(P, n_out, generator=g).to(DEV)
    enc_d, feat_d = enc.to(DEV), feat.to(DEV)
    xs1, e1 = _split_rows(enc)
    xs2, e2 = _split_rows(feat)
    if fp16:
        xs1, xs2 = xs1[:, :80].contiguous(), xs2[:, :n2].contiguous()
    xs1, e1, xs2, e2 = xs1.to(DEV), e1.to(DEV), xs2.to(DEV), e2.to(DEV)
    r1, r2 = _rows(DnsSplitRows, xs1, e1, 80, lo=not fp16), _rows(DnsSplitRows, xs2, e2, n2, lo=not fp16)
    flag = ops.MLP_FP16_FLAG if fp16 else 0
    y = [torch.zeros(P, n_out, device=DEV) for _ in range(2)]
    dx1 = [torch.zeros(P, 80, device=DEV) for _ in range(2)]
    dx2 = [torch.zeros(P, n2, device=DEV) for _ in range(2)]
    dp = [torch.zeros_like(params) for _ in range(2)]
    ws = [torch.zeros(int(lib.dns_mlp_bwd_ws_floats(P, nn, nl)), device=DEV) for _ in range(2)]
    check(lib.dns_mlp_fwd(ptr(enc_d), 80, ptr(feat_d), n2, n1, ptr(params), n_in, n_out, nn, nl, ptr(y[0]), n_out, P, None, None, 0, None,
                          flag, stream_ptr()), "dns_mlp_fwd")
    check(lib.dns_mlp_fwd_split(C.byref(r1), C.byref(r2), n1, ptr(params), n_in, n_out, nn, nl, ptr(y[1]), n_out, P, None, None, 0, flag,
                                stream_ptr()), "dns_mlp_fwd_split")
    check(lib.dns_mlp_bwd(ptr(enc_d), 80, ptr(feat_d), n2, n1, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(dx1[0]), 80,
                          ptr(dx2[0]), n2, ptr(dp[0]), ptr(ws[0]), P, None, None, 0, None, flag | ops.MLP_NO_DWIN_FLAG, stream_ptr()),
          "dns_mlp_bwd")
    check(lib.dns_mlp_bwd_split(C.byref(r1), C.byref(r2), n1, ptr(dy), n_out, ptr(params), n_in, n_out, nn, nl, ptr(dx1[1]), 80,
                                ptr(dx2[1]), n2, ptr(dp[1]), ptr(ws[1]), P, None, None, 0, flag, stream_ptr()), "dns_mlp_bwd_split")
    torch.cuda.synchronize()
    tol = 2e-3 if fp16 else 1e-6
    for a, b, name in ((y[1], y[0], "y"), (dx1[1][:, :48], dx1[0][:, :48], "dX segment 1"), (dx2[1], dx2[0], "dX segment 2"),
                       (ws[1], ws[0], "dH_1"), (dp[1], dp[0], "dW")):
        assert_close(a.cpu(), b.cpu(), rtol=tol, elementwise=False, what=f"two-segment split rows: {name}")
    assert float(dx1[1][:, 48:].abs().max()) == 0.0           # columns of the first row set beyond n_in1 are not part of the input
    if not fp16:
        xo = torch.cat((enc[:, :48], feat), 1).double()
        yo = tr.mlp_forward(xo, params.cpu().double(), n_in, n_out, nn, nl)
        assert_close(y[1].cpu(), yo.float(), rtol=1e-4, elementwise=False, what="two-segment split rows vs oracle")


def test_mlp_split_rows_argument_checks():
    ops, lib, DnsSplitRows, check, ptr, stream_ptr = _lib()
    P = 64
    xs = torch.zeros(P, 160, device=DEV, dtype=torch.float16)
    ex = torch.zeros(P, device=DEV, dtype=torch.int32)
    params = torch.zeros(ops.mlp_param_count(80, 33, 32, 1), device=DEV)
    y = torch.zeros(P, 33, device=DEV)
    bad = DnsSplitRows(xs.data_ptr(), ex.data_ptr(), 160, 0)           # no lo plane without DNS_MLP_FP16
    with pytest.raises(ValueError):
        check(lib.dns_mlp_fwd_split(C.byref(bad), None, 0, ptr(params), 80, 33, 32, 1, ptr(y), 33, P, None, None, 0, 0, stream_ptr()), "x")
    ok = DnsSplitRows(xs.data_ptr(), ex.data_ptr(), 160, 80)
    with pytest.raises(ValueError):                                     # n_in % 16 != 0
        check(lib.dns_mlp_fwd_split(C.byref(ok), None, 0, ptr(params), 72, 33, 32, 1, ptr(y), 33, P, None, None, 0, 0, stream_ptr()), "x")
    check(lib.dns_mlp_fwd_split(C.byref(ok), None, 0, ptr(params), 80, 33, 32, 1, ptr(y), 33, P, None, None, 0, 0, stream_ptr()), "x")
