"""dns_mlp_bwd without weight gradients (the frozen-scene form: recompute + dH chain + dX) per launch.  Round 4 measured this
form with 4 waves per workgroup (ONE per SIMD: 96.5 / 119.2 / 90.2 / 66.4 us for the four shapes below) against 8 waves per
workgroup (TWO per SIMD on one set of LDS weight images: 75.4 / 91.3 / 66.0 / 49.2 us, bit-identical dX) -- what latency hiding
alone is worth for the backward kernel; the 8-wave form is what the library now launches whenever no weight gradients are asked for."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops
from dns_slam_amd._lib import check, ptr, stream_ptr, LIB_PATH

lib = ops.lib._raw
P = int(os.environ.get("DNS_P", 262144))
dev = "cuda"
print("library:", LIB_PATH)
for n_in, n_out, nn, nl, two in ((80, 33, 64, 2, False), (112, 3, 64, 2, True), (80, 1, 64, 2, False), (80, 33, 32, 1, False)):
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(ops.mlp_param_count(n_in, n_out, nn, nl), generator=g) * 0.1).to(dev)
    enc = torch.randn(P, 80, generator=g).to(dev)
    feat = torch.randn(P, 64, generator=g).to(dev)
    dy = torch.randn(P, n_out, generator=g).to(dev)
    dx, dx2 = torch.empty(P, 80, device=dev), torch.empty(P, 64, device=dev)
    x2, n1 = (feat, 48) if two else (None, 0)
    st = stream_ptr()
    fn = lambda: check(lib.dns_mlp_bwd(ptr(enc), 80, ptr(x2), 64, n1, ptr(dy), n_out, ptr(w), n_in, n_out, nn, nl, ptr(dx), 80,
                                       ptr(dx2) if two else None, 64, None, None, P, None, None, 0, None, 0, st), "b")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{n_in}->{nn}x{nl}->{n_out}: backward without weight gradients {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch, "
          f"checksum {float(dx.double().abs().sum()):.6e}", flush=True)
