import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import ops
P = 262144
m = ops.GridMeta(16, 592)
tab = (torch.rand(m.total_rows * 2, device="cuda") * 2e-1 - 1e-1).requires_grad_(True)
pts = torch.rand(P, 3, device="cuda")
for _ in range(3):
    y = ops.encode(pts, tab, m, None, 16, True, True)
torch.cuda.synchronize()
ops.timer.arm()
for _ in range(10):
    y = ops.encode(pts, tab, m, None, 16, True, True)
torch.cuda.synchronize()
r = ops.timer.disarm()
print(", ".join(f"{k} {v[1] / v[0] * 1e3:.1f} us" for k, v in r.items()))
