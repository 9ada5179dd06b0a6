"""dns_encode_bwd with the point gradient (pose optimisation) on ray-ordered points: per-kernel event times."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import ops
P = 262144
m = ops.GridMeta(16, 592)
g = torch.Generator().manual_seed(0)
o = torch.rand(P // 64, 1, 3, generator=g) * 0.3 + 0.35
d = torch.randn(P // 64, 1, 3, generator=g) * 0.3
t = torch.linspace(0, 1, 64)[None, :, None]
pts = (o + d * t).reshape(-1, 3).clamp(0, 1).cuda().requires_grad_(True)
tab = (torch.rand(m.total_rows * 2, device="cuda") * 2e-1 - 1e-1)
y = ops.encode(pts, tab, m, None, 16, True, True)
gy = torch.randn_like(y)
for _ in range(3):
    pts.grad = None; y.backward(gy, retain_graph=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    pts.grad = None; y.backward(gy, retain_graph=True)
e1.record(); torch.cuda.synchronize()
print(f"encode backward (d_x only, table frozen): {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
