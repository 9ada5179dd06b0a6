"""dns_mlp_bwd WITH weight gradients (NO_DWIN) per launch, 262 144 points, five network shapes, with checksums of dX / dW / dH_1.
Round 4 used it for the producer / consumer backward (git b726e51, reverted: 8 waves per workgroup, the data-path waves publish
their K = point operands through LDS rings to accumulator waves on the same SIMD -- bit-equal checksums, 143 -> 188 us)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops
from dns_slam_amd._lib import check, ptr, stream_ptr

lib = ops.lib._raw
P = int(os.environ.get("DNS_P", 262144))
dev = "cuda"
for n_in, n_out, nn, nl, two, live in ((80, 33, 64, 2, False, 0), (112, 3, 64, 2, True, 80), (112, 8, 64, 2, True, 0), (80, 1, 64, 2, False, 0),
                                       (80, 33, 32, 1, False, 0)):
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(ops.mlp_param_count(n_in, n_out, nn, nl), generator=g) * 0.1).to(dev)
    enc = torch.randn(P, 80, generator=g).to(dev)
    feat = torch.randn(P, 64, generator=g).to(dev)
    dy = torch.randn(P, n_out, generator=g).to(dev)
    dx, dx2, dp = torch.zeros(P, 80, device=dev), torch.zeros(P, 64, device=dev), torch.zeros_like(w)
    ws = torch.empty(P * nn, device=dev)
    x2, n1 = (feat, 48) if two else (None, 0)
    st = stream_ptr()
    flag = ops.MLP_NO_DWIN_FLAG | (ops.MLP_LIVE_IN(live) if live else 0) | (3 if os.environ.get("DNS_ACC") else 0)   # DNS_ACC=1: read-add-write of d_x / d_x2
    fn = lambda: check(lib.dns_mlp_bwd(ptr(enc), 80, ptr(x2), 64, n1, ptr(dy), n_out, ptr(w), n_in, n_out, nn, nl, ptr(dx), 80,
                                       ptr(dx2) if two else None, 64, ptr(dp), ptr(ws), P, None, None, 0, None, flag, st), "b")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dp.zero_()
    fn()
    torch.cuda.synchronize()
    chk = (float(dx.double().abs().sum()), float(dp.double().abs().sum()), float(ws.double().abs().sum()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{n_in}{'(live %d)' % live if live else ''}->{nn}x{nl}->{n_out}: backward with weight gradients {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch, "
          f"checksums dX {chk[0]:.6e} dW {chk[1]:.6e} dH1 {chk[2]:.6e}", flush=True)
