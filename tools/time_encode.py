import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import ops
P = 262144
m = ops.GridMeta(16, 592)
tab = (torch.rand(m.total_rows * 2, device="cuda") * 2e-1 - 1e-1).requires_grad_(True)
if len(sys.argv) > 1 and sys.argv[1] == "rays":          # 4096 rays x 64 samples through the unit cube (the bench's access pattern)
    g = torch.Generator().manual_seed(0)
    o = torch.rand(P // 64, 1, 3, generator=g) * 0.3 + 0.35
    d = torch.randn(P // 64, 1, 3, generator=g) * 0.3
    pts = (o + d * torch.linspace(0, 1, 64)[None, :, None]).reshape(-1, 3).clamp(0, 1).cuda()
else:
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
