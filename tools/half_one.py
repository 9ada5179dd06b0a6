"""A few launches of dns_mlp_bwd_half (all gradients) / dns_mlp_fwd_half at one shape, for counter passes:
tools/sq_counters2.sh gpurun_out/sqh -- python3 tools/half_one.py;  python tools/sq_summary.py gpurun_out/sqh mlp_half"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops
P = int(os.environ.get("DNS_P", 1048576))
n_in, n_out, nn, nl = 80, 33, 64, 2
w = torch.randn(ops.mlp_param_count(n_in, n_out, nn, nl), device="cuda") * 0.1
x16 = torch.randn(P, n_in, device="cuda").half()
dy = torch.randn(P, n_out, device="cuda") * 1e-3
dx, dw, y = torch.zeros(P, n_in, device="cuda"), torch.zeros_like(w), torch.empty(P, n_out, device="cuda")
for _ in range(4):
    ops.mlp_fwd_half(x16, w, n_in, n_out, nn, nl, out=y)
    ops.mlp_bwd_half(x16, dy, w, n_in, n_out, nn, nl, d_x=dx, d_params=dw)
torch.cuda.synchronize()
