"""dns_mlp_dwin alone (event pairs, 20 launches).  Round 4 tried two half-row workgroups per point range on one XCD (48-64 instead
of 96-128 accumulator registers: 3-4 waves per SIMD instead of 2): 43.7 -> 41.3 us at 80 x 64, 50.7 -> 52.8 at 112 x 64, the step
1.87 -> 1.90 ms (the second read of the x rows costs what the occupancy gains) -- not kept."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops
from dns_slam_amd._lib import check, ptr, stream_ptr

lib = ops.lib._raw
P = int(os.environ.get("DNS_P", 262144))
dev = "cuda"
for n_in, nn, two in ((80, 64, False), (112, 64, True), (80, 32, False)):
    g = torch.Generator().manual_seed(0)
    enc = torch.randn(P, 80, generator=g).to(dev)
    feat = torch.randn(P, 64, generator=g).to(dev)
    ws = torch.randn(P * nn, generator=g).to(dev)
    dp = torch.zeros(ops.mlp_param_count(n_in, 8, nn, 2), device=dev)
    x2, n1 = (feat, 48) if two else (None, 0)
    st = stream_ptr()
    fn = lambda: check(lib.dns_mlp_dwin(ptr(enc), 80, ptr(x2), 64, n1, n_in, nn, 2, ptr(dp), ptr(ws), P, None, None, 0, 0, st), "d")
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dp.zero_()
    fn()
    torch.cuda.synchronize()
    chk = float(dp.double().abs().sum())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"dW_in {n_in} x {nn}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch, checksum {chk:.6e}", flush=True)
