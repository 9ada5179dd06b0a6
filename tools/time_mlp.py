"""Time dns_mlp_fwd / dns_mlp_bwd alone (event pairs), e.g. for DNS_GEMM_BLOCKS sweeps."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dns_slam_amd import ops
P = int(os.environ.get("DNS_P", 262144))
n_in, n_out, nn, nl = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (112, 8, 64, 2))]
dev = "cuda"
x = torch.randn(P, n_in, device=dev, requires_grad=True)
w = (torch.randn(ops.mlp_param_count(n_in, n_out, nn, nl), device=dev) * 0.1).requires_grad_(os.environ.get("DNS_NO_DW") is None)
gy = torch.randn(P, n_out, device=dev)
for _ in range(3):
    y = ops.mlp(x, w, n_in, n_out, nn, nl); y.backward(gy)
torch.cuda.synchronize()
ops.timer.arm()
for _ in range(10):
    y = ops.mlp(x, w, n_in, n_out, nn, nl); y.backward(gy)
torch.cuda.synchronize()
r = ops.timer.disarm()
print(f"{n_in}->{nn}x{nl}->{n_out}: " + ", ".join(f"{k} {v[1] / v[0] * 1e3:.1f} us" for k, v in r.items()))
