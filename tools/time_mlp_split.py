"""dns_mlp_fwd / dns_mlp_bwd on fp32 rows against dns_mlp_fwd_split / dns_mlp_bwd_split on split rows (event pairs, 20 launches)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops
from dns_slam_amd._lib import DnsSplitRows, check, ptr, stream_ptr

lib = ops.lib._raw
P = int(os.environ.get("DNS_P", 262144))
dev = "cuda"


def split_rows(x):
    m = x.abs().amax(dim=1)
    _, ex = torch.frexp(m)
    e = torch.where(m > 0, (14 - ex).clamp(-110, 110), torch.zeros_like(ex)).to(torch.int32)
    xs = torch.ldexp(x, e[:, None])
    hi = xs.half()
    lo = (xs - hi.float()).half()
    return torch.cat((hi, lo), 1).contiguous(), e.contiguous()


ONCE = os.environ.get("DNS_ONCE") is not None          # counter passes: two launches per variant, first shape only


def timed(fn, n=20):
    if ONCE:
        fn()
        fn()
        torch.cuda.synchronize()
        return 0.0
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SHAPES = ((80, 33, 64, 2, False), (112, 3, 64, 2, True), (112, 8, 64, 2, True), (80, 1, 64, 2, False), (80, 33, 32, 1, False))
if ONCE:
    SHAPES = SHAPES[int(os.environ["DNS_ONCE"]):][:1]
for n_in, n_out, nn, nl, two in SHAPES:
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(ops.mlp_param_count(n_in, n_out, nn, nl), generator=g) * 0.1).to(dev)
    enc = torch.randn(P, 80, generator=g).to(dev)
    feat = torch.randn(P, 64, generator=g).to(dev)
    dy = torch.randn(P, n_out, generator=g).to(dev)
    xs1, e1 = split_rows(enc)
    xs2, e2 = split_rows(feat)
    r1 = DnsSplitRows(xs1.data_ptr(), e1.data_ptr(), 160, 80)
    r2 = DnsSplitRows(xs2.data_ptr(), e2.data_ptr(), 128, 64)
    y = torch.empty(P, n_out, device=dev)
    dx, dx2, dp = torch.empty(P, 80, device=dev), torch.empty(P, 64, device=dev), torch.zeros_like(w)
    ws = torch.empty(P * nn, device=dev)
    x2, n1 = (feat, 48) if two else (None, 0)
    st = stream_ptr()
    res = {}
    for flag, tag in (((0, ""),) if ONCE else ((0, ""), (ops.MLP_FP16_FLAG, " fp16"))):
        res["fwd" + tag] = timed(lambda: check(lib.dns_mlp_fwd(ptr(enc), 80, ptr(x2), 64, n1, ptr(w), n_in, n_out, nn, nl, ptr(y), n_out, P, None, None, 0, None, flag, st), "f"))
        res["fwd split" + tag] = timed(lambda: check(lib.dns_mlp_fwd_split(C.byref(r1), C.byref(r2) if two else None, n1, ptr(w), n_in, n_out, nn, nl, ptr(y), n_out, P, None, None, 0, flag, st), "fs"))
        res["bwd" + tag] = timed(lambda: check(lib.dns_mlp_bwd(ptr(enc), 80, ptr(x2), 64, n1, ptr(dy), n_out, ptr(w), n_in, n_out, nn, nl, ptr(dx), 80, ptr(dx2) if two else None, 64, ptr(dp), ptr(ws), P, None, None, 0, None, flag | ops.MLP_NO_DWIN_FLAG, st), "b"))
        res["bwd split" + tag] = timed(lambda: check(lib.dns_mlp_bwd_split(C.byref(r1), C.byref(r2) if two else None, n1, ptr(dy), n_out, ptr(w), n_in, n_out, nn, nl, ptr(dx), 80, ptr(dx2) if two else None, 64, ptr(dp), ptr(ws), P, None, None, 0, flag, st), "bs"))
    print(f"{n_in}->{nn}x{nl}->{n_out}: " + ", ".join(f"{k} {v:.1f}" for k, v in res.items()) + "  (us per launch)", flush=True)
