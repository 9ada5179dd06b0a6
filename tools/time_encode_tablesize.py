import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from dns_slam_amd import ops
P = 262144
g = torch.Generator().manual_seed(0)
o = torch.rand(P // 64, 1, 3, generator=g) * 0.3 + 0.35
d = torch.randn(P // 64, 1, 3, generator=g) * 0.3
t = torch.linspace(0, 1, 64)[None, :, None]
pts = (o + d * t).reshape(-1, 3).clamp(0, 1).cuda()
for hs in (8, 12, 14, 16, 20):
    m = ops.GridMeta(hs, 592)
    tab = (torch.rand(m.total_rows * 2, device="cuda") * 2e-1 - 1e-1)
    for _ in range(3):
        y = ops.encode(pts, tab, m, None, 16, False, True)
    torch.cuda.synchronize()
    ops.timer.arm()
    for _ in range(10):
        y = ops.encode(pts, tab, m, None, 16, False, True)
    torch.cuda.synchronize()
    r = ops.timer.disarm()
    print(f"hash 2^{hs} ({m.total_rows * 8 / 1e6:.1f} MB): " + ", ".join(f"{k} {v[1] / v[0] * 1e3:.1f} us" for k, v in r.items()))
